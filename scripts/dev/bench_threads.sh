#!/bin/bash
# dev: the headline step against the size of the host thread pool (A/B pairs in one call)
for t in ${@:-32 64 32 64}; do
  DH_HOST_THREADS=$t python bench.py --steps 10 --warmup 3 --ref-steps 0 --no-cpu-baseline > /tmp/bt.json
  python - "$t" <<'PY'
import json, sys
j = json.loads(open('/tmp/bt.json').read().strip().splitlines()[-1])
print('threads', sys.argv[1], round(j["ms_per_step"], 1), {k: round(v, 1) for k, v in j["stages_ms"].items() if k in ("map_wall", "collect_wall", "process_wall")})
PY
done
