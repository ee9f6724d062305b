for pb in 27 28 29; do
DH_INDEX_PBITS=$pb DH_TRACE=1 timeout -s KILL 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_t.json 2> gpurun_out/bench_t.err
echo "pbits $pb"; grep "dh_align_db\] A=60000" gpurun_out/bench_t.err | tail -1 | cut -c130-330
grep "dh_align_db\] A=1001" gpurun_out/bench_t.err | tail -1 | cut -c150-330
done
