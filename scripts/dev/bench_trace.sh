DH_TRACE=1 timeout -s KILL 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_t.json 2> gpurun_out/bench_t.err
grep -v "dh_align_db\]" gpurun_out/bench_t.err | tail -30
