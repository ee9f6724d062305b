timeout -s KILL 300 python -m pytest tests/test_parity_map_gpu.py -x -q 2>&1 | tail -3
for cfg in "30 30" "14 30" "14 14" "14 20"; do timeout -s KILL 120 python scripts/dev/wave_width_bench.py $cfg 2>&1 | tail -1; done
