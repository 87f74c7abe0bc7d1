#!/bin/bash
# kernel-trace pass only: tools/prof_stats.sh <tag> [cfg2|cfg3]  -> gpurun_out/profiles_out/<tag>[_cfg3]_kernel_stats.csv
set -e
tag=$1; cfg=${2:-cfg2}
steps=30; [ "$cfg" = cfg3 ] && steps=5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
base=$R/gpurun_out/prof_${tag}_${cfg}
rm -rf ${base}_stats && mkdir -p ${base}_stats
rocprofv3 --kernel-trace --stats -d ${base}_stats --output-format csv -- python3 $R/bench.py --config $cfg --steps $steps --warmup 3 --no-cpu > ${base}_stats/bench.json 2> ${base}_stats/bench.err
cd $R
sfx=""; [ "$cfg" != cfg2 ] && sfx="_$cfg"
mkdir -p gpurun_out/profiles_out
python profiles/summarize.py stats ${base}_stats gpurun_out/profiles_out/${tag}${sfx}_kernel_stats.csv
cp ${base}_stats/bench.json gpurun_out/profiles_out/${tag}${sfx}_bench_under_profiler.json
