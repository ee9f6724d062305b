timeout -s KILL 400 python scripts/dev/pile_seed.py 8192 4096 2048 2>&1 | tail -3
