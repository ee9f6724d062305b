timeout -s KILL 600 python -m pytest tests/test_parity_shard_gpu.py tests/test_parity_map_gpu.py -x -q 2>&1 | tail -3
timeout -s KILL 600 python scripts/dev/shard_glue.py 2>&1 | grep -E "direct|pack_candidates|merge_candidates|pile_costs|arrays|process\)|select|from_flat|create" | head -14
