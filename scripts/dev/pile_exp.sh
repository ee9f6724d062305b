#!/bin/bash
# dev: the uncapped pile-up stage under variants of the tile launch (phase clocks: build_prof.sh)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
run() { echo "=== $*"; env "$@" DH_DEV_LIB=scripts/dev/libdentist_hip_prof.so DH_TRACE=1 python scripts/dev/pile_uncapped.py 2 2>&1 | grep -E "tile prof|join prof|seed prof|^process |A=165927|A=1[0-9]+ seqs.*B=1[0-9]+ seqs" | grep -v "A=1000 " | cut -c1-420 | head -${NL:-9}; }
for v in "$@"; do run $(echo $v | tr ',' ' '); done
