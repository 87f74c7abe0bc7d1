#!/bin/bash
# cfg3 bench line only (no CPU legs): tools/bench_cfg3.sh <tag>
tag=${1:-r03}
python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu > gpurun_out/${tag}_cfg3_bench.json 2> gpurun_out/${tag}_cfg3_bench.err
echo cfg3 rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_cfg3_bench.json"))
print("cfg3 ms_per_step", d["ms_per_step"], d["stage_ms"])
PY
