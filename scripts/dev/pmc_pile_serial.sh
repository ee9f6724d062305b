#!/bin/bash
# dev: SQ counters of the uncapped process stage with its parts one after the other (no two kernels share the device, so the
# counters of a launch are its own)
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$root"
rm -rf /tmp/pps1 /tmp/pps2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/pps1 -o run -- python scripts/dev/pile_uncapped.py 1 > /tmp/pps1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d /tmp/pps2 -o run -- python scripts/dev/pile_uncapped.py 1 > /tmp/pps2.log 2>&1
python "$root/scripts/pmc_summary.py" /tmp/pps1 /tmp/pps2 | grep -E "k_tile|k_join<|k_seed<8192|k_seed<16384|k_seg_vote2" | cut -c1-260
