#!/bin/bash
# dev: bench.py as EIGHT processes over gloo on ONE GPU (--dev-share-gpu): the multi-process path at world 8 -- results, not times
export MASTER_ADDR=127.0.0.1
timeout -s KILL 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 8 --steps 2 --warmup 1 --dev-share-gpu --no-cpu-baseline "$@" > gpurun_out/bench_n8.json 2> gpurun_out/bench_n8.err
python - <<'PY'
import json
lines = [l for l in open('gpurun_out/bench_n8.json').read().strip().splitlines() if l.startswith('{')]
d = json.loads(lines[-1])
print('n_gpus', d['n_gpus'], 'ms_per_step %.1f value %.0f err %.5f closed %d' % (d['ms_per_step'], d['value'], d['config']['consensus_error_rate'], d['config']['gaps_closed']), d['config']['collectives'], 'edits', d['config']['consensus_edit_distance_vs_truth'])
PY
tail -3 gpurun_out/bench_n8.err
