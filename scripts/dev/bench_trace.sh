DH_TRACE=1 timeout -s KILL 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_t.json 2> gpurun_out/bench_t.err
grep "dh_align_db\]" gpurun_out/bench_t.err | tail -6 | cut -c1-420
grep "dh_process\]" gpurun_out/bench_t.err | tail -9
