export MASTER_ADDR=127.0.0.1
timeout -s KILL 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --dev-share-gpu --no-cpu-baseline "$@" > gpurun_out/bench_n2.json 2> gpurun_out/bench_n2.err
python - <<'PY'
import json
lines = [l for l in open('gpurun_out/bench_n2.json').read().strip().splitlines() if l.startswith('{')]
d = json.loads(lines[-1])
print('n_gpus', d['n_gpus'], 'ms_per_step %.1f value %.0f err %.5f closed %d' % (d['ms_per_step'], d['value'], d['config']['consensus_error_rate'], d['config']['gaps_closed']))
print({k: round(v, 1) for k, v in d['stages_ms'].items()})
PY
tail -5 gpurun_out/bench_n2.err
