#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: tools/profile_round.sh <tag> [cfg2|cfg3]   (on the GPU box)
# separate runs: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; --pmc <SQ counters>
set -e
tag=$1; cfg=${2:-cfg2}
steps=30; [ "$cfg" = cfg3 ] && steps=5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
base=$R/gpurun_out/prof_${tag}_${cfg}
rm -rf ${base}_* && mkdir -p ${base}_stats ${base}_fetch ${base}_write ${base}_sq
rocprofv3 --kernel-trace --stats -d ${base}_stats --output-format csv -- python3 $R/bench.py --config $cfg --steps $steps --warmup 3 --no-cpu > ${base}_stats/bench.json 2> ${base}_stats/bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${base}_fetch --output-format csv -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu > /dev/null 2> ${base}_fetch/bench.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ${base}_write --output-format csv -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu > /dev/null 2> ${base}_write/bench.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d ${base}_sq --output-format csv -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu > /dev/null 2> ${base}_sq/bench.err
cd $R
sfx=""; [ "$cfg" != cfg2 ] && sfx="_$cfg"
python profiles/summarize.py stats ${base}_stats profiles/${tag}${sfx}_kernel_stats.csv
python profiles/summarize.py pmc ${base}_fetch ${base}_write $cfg profiles/${tag}${sfx}_pmc_summary.csv
python profiles/summarize.py sq ${base}_sq profiles/${tag}${sfx}_sq_summary.csv $cfg
cp ${base}_stats/bench.json profiles/${tag}${sfx}_bench_under_profiler.json
mkdir -p gpurun_out/profiles_out && cp profiles/${tag}${sfx}_* profiles/pmc_traffic.json gpurun_out/profiles_out/
