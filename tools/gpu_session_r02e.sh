#!/bin/bash
# Round 2, session E (evidence): rocprofv3 kernel-trace + FETCH_SIZE / WRITE_SIZE passes per BASELINE config,
# VALU issue share of the metric kernels, the bench line.  Usage on the GPU box: bash tools/gpu_session_r02e.sh
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT/r02e
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/r02e/pytest_gpu.log
echo "== bench.py"
timeout 600 python bench.py 2>&1 | tail -1 | tee $OUT/r02e/bench_1gpu.json
prof() {  # tag, command...
  local tag=$1; shift
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$tag -o stats -- "$@" > $OUT/prof_stats_$tag.log 2>&1
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch_$tag -o fetch -- "$@" > $OUT/prof_fetch_$tag.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write_$tag -o write -- "$@" > $OUT/prof_write_$tag.log 2>&1
  cd $REPO
  python tools/summarize_prof.py $OUT $tag > $OUT/r02e/rocprof_summary_$tag.txt 2>&1
  cp $OUT/pmc_traffic_$tag.json $OUT/r02e/ 2>/dev/null
  echo "-- $tag"; head -12 $OUT/r02e/rocprof_summary_$tag.txt
}
prof r02_bench python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline
prof r02_cfg3 python $REPO/tools/bench_configs.py --configs 3 --reps 5
prof r02_cfg4 python $REPO/tools/bench_configs.py --gpus 1 --configs 4 --records 16 --batch-records 8
prof r02_cfg5 python $REPO/tools/bench_configs.py --gpus 1 --configs 5 --reps 5
echo "== VALU issue share (config 3 kernels + the bare stencils)"
cd /tmp
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $OUT/prof_valu_r02 -o valu -- python $REPO/tools/microbench.py --reps 5 --cases stencil,metric,cumsum,reduce > $OUT/prof_valu_r02.log 2>&1
cd $REPO
python tools/valu_util.py $(find $OUT/prof_valu_r02 -name "*.db" | head -1) 2>&1 | tee $OUT/r02e/valu_issue_share.txt
echo "== kernel table (microbench, all kernels)"
timeout 300 python tools/microbench.py --reps 9 --cases copy,stencil,metric,cumsum,reduce,vort,generic 2>/dev/null | grep '^{' > $OUT/r02e/microbench_all_kernels.jsonl
python - <<'PY'
import json
for ln in open("gpurun_out/r02e/microbench_all_kernels.jsonl"):
    r = json.loads(ln)
    print(f"{r['case']:55s} {r['ms']:7.3f} ms {r['frac_8TBps']*100:5.1f}%")
PY
