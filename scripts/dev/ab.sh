#!/bin/bash
# dev: A/B of environment knobs on the default bench line: ab.sh "NAME=VAL ..." ...   (each argument one variant; "" = baseline)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
mkdir -p gpurun_out/ab
i=0
for v in "$@"; do
  i=$((i+1))
  for rep in 1 2; do
    env $v python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('%-40s ms/step %.1f  map %.1f collect %.1f process %.1f' % ('$v' or 'baseline', d['ms_per_step'], s['map_wall'], s['collect_wall'], s['process_wall']))"
  done
done
