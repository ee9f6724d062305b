#!/bin/bash
# dev: the library with the phase clocks of the seed / join / tile kernels compiled in (-DDH_SEED_PROF) ->
# scripts/dev/libdentist_hip_prof.so (git-ignored; travels to the GPU box; select it with DH_DEV_LIB)
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/prof
for f in dentist_amd/csrc/*.hip dentist_amd/csrc/*.cpp; do
  o=build/prof/$(basename $f).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find dentist_amd/csrc include -name '*.h' -newer $o)" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DDH_SEED_PROF ${EXTRA_DEFS:-} -c -o $o $f &
  fi
done
wait
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o scripts/dev/libdentist_hip_prof.so build/prof/*.o
ls -la scripts/dev/libdentist_hip_prof.so
