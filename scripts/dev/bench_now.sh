timeout -s KILL 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
tail -c 3000 gpurun_out/bench_now.json
tail -3 gpurun_out/bench_now.err
