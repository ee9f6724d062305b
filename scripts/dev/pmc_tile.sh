#!/bin/bash
# dev: memory-side counters of the k_tile launches of one bench step (separate --pmc passes)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_TAG_STALL_sum TCC_REQ_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pt_$i -o run -- python "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --workload cfg1_10Mb_100gaps_100kx10kb "$@" > /tmp/pt_$i.log 2>&1
  python "$root/scripts/pmc_summary.py" /tmp/pt_$i | grep -E " k_tile " | cut -c1-200; tail -2 /tmp/pt_$i.log | cut -c1-200
done
