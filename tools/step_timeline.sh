#!/bin/bash
# kernel timeline of one steady-state headline step (timing level 0): tools/step_timeline.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_timeline
rm -rf $out && mkdir -p $out
TIMING=0 rocprofv3 --kernel-trace -d $out --output-format csv -- python3 $R/tools/headline_frame.py stepwise > $out/run.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$out/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_bounds<true>" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
prev = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:44]:44s} dur {(e - s) / 1e3:7.1f} gap {((s - prev) / 1e3 if prev else 0):6.1f}")
    prev = e
print("step span", (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
PY
