for cfg in "30 120" "14 60"; do set -- $cfg; timeout -s KILL 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --map-width $1 --map-xdrop $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=j['config']; s=j['stages_ms']
print(c['mapping_width'], c['mapping_xdrop'], 'ms/step %.1f'%j['ms_per_step'], 'closed', c['gaps_closed'], 'gapbp', c['gap_bases_closed'], 'err %.5f'%c['consensus_error_rate'], 'map_wave %.1f map_seed %.1f all_wave %.1f'%(s['map_wave'], s['map_seed'], s['all_wave']), 'value %.0f'%j['value'])
"; done
