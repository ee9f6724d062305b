timeout -s KILL 600 python bench.py --workload cfg1_10Mb_100gaps_100kx10kb --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg1.json 2> gpurun_out/bench_cfg1.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_cfg1.json').read().strip().splitlines()[-1])
print('ms_per_step %.1f value %.0f err %.5f closed %d' % (d['ms_per_step'], d['value'], d['config']['consensus_error_rate'], d['config']['gaps_closed']))
print({k: round(v, 1) for k, v in d['stages_ms'].items()})
print('cells/s %.3g' % d['roofline']['wave_cells_per_s'], 'wave ms', d['roofline']['kernel_ms_per_step'])
PY
tail -2 gpurun_out/bench_cfg1.err
