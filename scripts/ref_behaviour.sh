#!/bin/bash
# The headline configuration of bench.py uses a read cap (60 entries per pile-up) and modimer sampling (1/8) that the
# reference does not apply.  This measures the same workload WITHOUT them and writes profiles/reference_behaviour.json,
# which bench.py attaches to its line as "reference_behaviour".  Run on the GPU box:  bash scripts/ref_behaviour.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --max-reads 0 > $out/rb_uncapped.json 2> $out/rb_uncapped.err
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --kmer-mod 1 > $out/rb_mod1.json 2> $out/rb_mod1.err
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --kmer-mod 1 --max-reads 0 > $out/rb_both.json 2> $out/rb_both.err
python - <<'PY'
import json, os, sys
sys.path.insert(0, "scripts")
from traffic_json import kernel_build_id
res = {"kernel_build_id": kernel_build_id(),
       "source": "scripts/ref_behaviour.sh: python bench.py --steps 2 --warmup 1 --no-cpu-baseline <flags>, one MI355X"}
for key, name in (("max_reads_0", "rb_uncapped"), ("kmer_mod_1", "rb_mod1"), ("max_reads_0_kmer_mod_1", "rb_both")):
    try:
        d = json.loads(open(f"gpurun_out/{name}.json").read().strip().splitlines()[-1])
        res[key] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"],
                    "pile_up_entries": d["config"]["pile_up_entries"], "gaps_closed": d["config"]["gaps_closed"],
                    "consensus_error_rate": d["config"]["consensus_error_rate"],
                    "stages_ms": {k: d["stages_ms"][k] for k in ("map_wall", "collect_wall", "process_wall")}}
    except Exception as e:  # noqa: BLE001
        res[key] = {"error": str(e)}
json.dump(res, open("gpurun_out/reference_behaviour.json", "w"), indent=1)
print(json.dumps(res))
PY
