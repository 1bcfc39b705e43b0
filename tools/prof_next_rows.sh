#!/usr/bin/env bash
# PMC evidence for the "next" rows: HBM bytes per launch of the complex-topology and transform kernels.
# usage: bash tools/prof_next_rows.sh <tag>   (on the GPU box; results under gpurun_out/)
set -u
TAG=${1:-r01n}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $REPO/tools/bench_configs.py --reps 3 --configs f2,f4"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$TAG -o stats -- $CMD > $OUT/prof_stats_$TAG.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch_$TAG -o fetch -- $CMD > $OUT/prof_fetch_$TAG.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write_$TAG -o write -- $CMD > $OUT/prof_write_$TAG.log 2>&1
cd $REPO
python tools/summarize_prof.py $OUT $TAG 2>&1 | tee $OUT/prof_summary_$TAG.txt
