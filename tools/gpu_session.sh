#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 stats + PMC passes, tuning sweeps.
# Usage (from the repo root on the GPU box): bash tools/gpu_session.sh <tag>
TAG=${1:-r01}
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu_$TAG.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke_$TAG.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -2 | tee $OUT/bench_$TAG.json
cd /tmp
echo "== rocprofv3 kernel-trace stats"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$TAG -o stats -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_stats_$TAG.log 2>&1
echo "== rocprofv3 pmc FETCH_SIZE"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch_$TAG -o fetch -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch_$TAG.log 2>&1
echo "== rocprofv3 pmc WRITE_SIZE"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write_$TAG -o write -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_write_$TAG.log 2>&1
cd $REPO
find $OUT/prof_stats_$TAG $OUT/prof_fetch_$TAG $OUT/prof_write_$TAG -type f | head -30
python tools/summarize_prof.py $OUT $TAG 2>&1 | tee $OUT/prof_summary_$TAG.txt
echo "== microbench (all kernels, random data)"
timeout 300 python tools/microbench.py --reps 9 --cases copy,stencil,metric,cumsum,reduce,vort,generic > $OUT/mb_full_$TAG.jsonl 2>&1
grep -v amdgpu.ids $OUT/mb_full_$TAG.jsonl
