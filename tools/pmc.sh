#!/bin/bash
# counter pass of the headline bench: tools/pmc.sh <tag> "<COUNTER ...>" [extra bench args]
# (own rocprofv3 run with --kernel-trace only, as the pool requires); prints per-kernel means
set -e
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --pmc $ctrs -d $out --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $out/bench.json 2> $out/bench.err
cd $GRAFT_REPO_ROOT && python - "$out" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    a = acc[r["Kernel_Name"]][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(acc):
    if "ag2" not in k: continue
    print(k[:60], {c: round(v[1] / v[0], 1) for c, v in acc[k].items()}, "calls", max(v[0] for v in acc[k].values()))
PY
