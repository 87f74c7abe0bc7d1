#!/bin/bash
# on the GPU box: tools/ab_frontend.sh name1 name2 ...  -- the with_front_end leg under each library variant
# (base = the product build), with a kernel trace of the front-end kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for name in "$@"; do
  lib=$R/agile_grasp2_amd/csrc/exp/libag2hip_${name}.so
  [ "$name" = base ] && lib=$R/agile_grasp2_amd/csrc/libag2hip.so
  out=$R/gpurun_out/ab_fe_${name}
  rm -rf $out && mkdir -p $out
  AG2_LIB=$lib python3 $R/tools/front_end_leg.py 30 2>/dev/null | cut -c1-200
  AG2_LIB=$lib rocprofv3 --kernel-trace --stats -d $out --output-format csv -- python3 $R/tools/front_end_leg.py 30 > /dev/null 2>&1
  python3 - "$out" "$name" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(t in r["Name"] for t in ("k_vox_", "k_raw_filter", "k_sel_small", "k_scan_chained<true>")):
        print("  ", sys.argv[2], r["Name"].split("(")[0][:40], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
