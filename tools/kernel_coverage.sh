#!/bin/bash
# Which kernels of libag2hip.so does the GPU test suite launch?  (on the GPU box)
# tools/kernel_coverage.sh -> gpurun_out/kernel_coverage.txt: every __global__ name in csrc/*.hip with its calls
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_cov
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out --output-format csv -- python3 -m pytest $R/tests -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1
echo "pytest rc=$?" ; tail -2 $out/pytest.log
cd $R
python3 - <<PY
import csv, glob, re
calls = {}
for f in glob.glob("$out/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        m = re.search(r"ag2::(k_[a-z0-9_]+)", name)
        if m:
            calls[m.group(1)] = calls.get(m.group(1), 0) + int(r["Calls"])
names = set()
for f in glob.glob("agile_grasp2_amd/csrc/*.hip"):
    src = open(f).read()
    for m in re.finditer(r"__global__[^;{]*?\b(k_[a-z0-9_]+)\s*\(", src):
        names.add(m.group(1))
with open("gpurun_out/kernel_coverage.txt", "w") as o:
    for n in sorted(names):
        o.write(f"{n} {calls.get(n, 0)}\n")
print("kernels", len(names), "never launched:", [n for n in sorted(names) if calls.get(n, 0) == 0])
PY
