#!/bin/bash
# kernel-trace of any harness script: tools/prof_script.sh <tag> <script.py> [args...] -> gpurun_out/profiles_out/<tag>_kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out $R/gpurun_out/profiles_out
script=$R/$1; shift
rocprofv3 --kernel-trace --stats -d $out --output-format csv -- python3 $script "$@" > $out/stdout.txt 2> $out/stderr.txt
cd $R && python profiles/summarize.py stats $out gpurun_out/profiles_out/${tag}_kernel_stats.csv
