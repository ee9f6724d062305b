timeout -s KILL 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_q.json').read().strip().splitlines()[-1])
print('ms_per_step %.1f value %.0f err %.5f closed %d' % (d['ms_per_step'], d['value'], d['config']['consensus_error_rate'], d['config']['gaps_closed']))
print({k: round(v, 1) for k, v in d['stages_ms'].items()})
print('cells/s %.3g' % d['roofline']['wave_cells_per_s'])
PY
tail -2 gpurun_out/bench_q.err
