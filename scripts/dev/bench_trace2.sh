# dev: traced bench (2 steps) -> the per-call lines of the process stage and the stage table
OUT=${1:-r4_t}
DH_TRACE=1 timeout -s KILL 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$OUT.json 2> gpurun_out/$OUT.err
grep "dh_align_db\]" gpurun_out/$OUT.err | tail -12 | cut -c1-420
grep "dh_process\]" gpurun_out/$OUT.err | tail -20
python - <<PY
import json
d = json.loads(open('gpurun_out/$OUT.json').read().strip().splitlines()[-1])
print('ms_per_step %.1f value %.0f err %.5f closed %d' % (d['ms_per_step'], d['value'], d['config']['consensus_error_rate'], d['config']['gaps_closed']))
print({k: round(v, 1) for k, v in d['stages_ms'].items()})
PY
tail -3 gpurun_out/$OUT.err
