for n in 1 4 40; do timeout -s KILL 50 python scripts/dev/t_width.py $n 14 2>&1 | tail -3; echo "n=$n rc=$?"; done
DH_WAVE_G32=1 timeout -s KILL 50 python scripts/dev/t_width.py 40 14 2>&1 | tail -2
