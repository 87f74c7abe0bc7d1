#!/bin/bash
# kernel trace of the with_front_end leg: tools/prof_frontend.sh <tag>   (on the GPU box)
#   -> profiles/<tag>_frontend_kernel_stats.csv, profiles/<tag>_frontend_host.json (wall time per frame, the
#      kernels' sum per frame, the gap between the two)
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
base=$R/gpurun_out/prof_${tag}_frontend
rm -rf ${base}_stats && mkdir -p ${base}_stats
python3 $R/tools/front_end_leg.py 30 > ${base}_stats/plain.json 2> ${base}_stats/plain.err
rocprofv3 --kernel-trace --stats -d ${base}_stats --output-format csv -- python3 $R/tools/front_end_leg.py 30 > ${base}_stats/leg.json 2> ${base}_stats/leg.err
cd $R
python profiles/summarize.py stats ${base}_stats profiles/${tag}_frontend_kernel_stats.csv
python - "$tag" "${base}_stats" <<'PY'
import csv, json, sys
tag, d = sys.argv[1], sys.argv[2]
plain = json.loads(open(d + "/plain.json").read().strip().splitlines()[-1])
prof = json.loads(open(d + "/leg.json").read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(f"profiles/{tag}_frontend_kernel_stats.csv")))
frames = 33  # 30 timed + 3 warm-up (the first of them step by step)
per = {r["Name"].split("(")[0].replace("ag2::", "").replace("void ", ""): float(r["TotalDurationNs"]) / frames / 1e3 for r in rows}
front = {k: v for k, v in per.items() if any(t in k for t in ("k_raw_filter", "k_vox_", "k_sel_", "k_scan_chained<true>"))}
out = {"without_profiler": plain, "under_profiler": prof,
       "kernel_us_per_frame_total": round(sum(per.values()), 1),
       "front_end_kernels_us_per_frame": {k: round(v, 1) for k, v in front.items()},
       "front_end_kernels_us_per_frame_sum": round(sum(front.values()), 1),
       "host_gap_us_per_frame": round(plain["ms_per_frame_mean"] * 1e3 - sum(per.values()), 1),
       "note": "gap = wall time of one ag2_detect_frame_raw call (no profiler) minus the kernels' summed durations (kernel trace of the same script): launch of the pack kernel + graph launch + the one synchronisation + result copy"}
json.dump(out, open(f"profiles/{tag}_frontend_host.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
mkdir -p gpurun_out/profiles_out && cp profiles/${tag}_frontend_* gpurun_out/profiles_out/
