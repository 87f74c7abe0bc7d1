#!/bin/bash
# kernel trace of one leg of the cfg5 stream: bash tools/prof_cfg5.sh <tag> [graph|plain|stepwise]
set -e
tag=${1:-x}
export AG2_STREAM_LEG=${2:-graph}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg5_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 30 > $out/bench.json 2> $out/bench.err
cd $GRAFT_REPO_ROOT && python profiles/summarize.py stats $out gpurun_out/prof_cfg5_${tag}_stats.csv
