#!/bin/bash
# rocprofv3 kernel trace of the cfg5 graph leg (RAW frames through ag2_detect_frame_raw):
#   tools/prof_cfg5.sh <tag>      (on the GPU box)  -> profiles/<tag>_cfg5_kernel_stats.csv
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
base=$R/gpurun_out/prof_${tag}_cfg5
rm -rf ${base}_stats && mkdir -p ${base}_stats
AG2_STREAM_LEG=graph rocprofv3 --kernel-trace --stats -d ${base}_stats --output-format csv -- python3 $R/bench.py --config cfg5 --steps 30 > ${base}_stats/bench.json 2> ${base}_stats/bench.err
cd $R
python profiles/summarize.py stats ${base}_stats profiles/${tag}_cfg5_kernel_stats.csv
cp ${base}_stats/bench.json profiles/${tag}_cfg5_graph_leg_under_profiler.json
mkdir -p gpurun_out/profiles_out && cp profiles/${tag}_cfg5_* gpurun_out/profiles_out/
