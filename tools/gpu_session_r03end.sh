#!/bin/bash
# Round 3, session ZZ (final evidence of the round, last build: K2Sy, K2Sm, lean transforms, pad tiles, reduce specialisations): full GPU suite, smoke, bench line (plain and through RCCL with one rank), config
# tables, kernel tables (f64 / f32), operator survey, rocprofv3 kernel-trace + FETCH_SIZE / WRITE_SIZE passes per
# BASELINE config, VALU issue share.  Usage on the GPU box: bash tools/gpu_session_final.sh
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03end
mkdir -p $S
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee $S/pytest_gpu.log
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $S/smoke.log
echo "== bench"; timeout 400 python bench.py 2>&1 | tail -1 | tee $S/bench_1gpu.json
echo "== bench through RCCL, one rank"; XG_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee $S/bench_rccl_ws1.json
echo "== configs"; timeout 300 python tools/bench_configs.py --configs 2,3 2>&1 | grep '^{' | tee $S/configs_2_3.jsonl | cut -c1-220
timeout 400 python tools/bench_configs.py --gpus 1 --configs 4,5 --records 45 2>&1 | grep '^{' | tee $S/configs_4_5_sharded_1rank.jsonl | cut -c1-260
timeout 300 python tools/bench_configs.py --configs 5x,f1 2>&1 | grep '^{' > $S/configs_5x_f1.jsonl
echo "== kernel tables + survey"
timeout 200 python tools/microbench.py --reps 9 --cases copy,stencil,metric,cumsum,reduce,vort,generic 2>/dev/null | grep '^{' > $S/microbench_all_kernels.jsonl
timeout 200 python tools/microbench.py --reps 9 --dtype f32 --cases stencil,metric,cumsum,reduce,vort 2>/dev/null | grep '^{' > $S/microbench_f32.jsonl
timeout 200 python tools/survey.py --reps 7 2>&1 | grep '^{' > $S/survey.jsonl
prof() {  # tag, command...
  local tag=$1; shift
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$tag -o stats -- "$@" > $OUT/prof_stats_$tag.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch_$tag -o fetch -- "$@" > $OUT/prof_fetch_$tag.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write_$tag -o write -- "$@" > $OUT/prof_write_$tag.log 2>&1
  cd $REPO
  python tools/summarize_prof.py $OUT $tag > $S/rocprof_summary_$tag.txt 2>&1
  cp $OUT/pmc_traffic_$tag.json $S/ 2>/dev/null
  echo "-- $tag"; head -8 $S/rocprof_summary_$tag.txt
}
prof r03end_bench python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline
prof r03end_cfg3 python $REPO/tools/bench_configs.py --configs 3 --reps 5
prof r03end_cfg4 python $REPO/tools/bench_configs.py --gpus 1 --configs 4 --records 16 --batch-records 8
prof r03end_cfg5 python $REPO/tools/bench_configs.py --gpus 1 --configs 5 --reps 5
prof r03end_marches python $REPO/tools/ab_tunables.py --cases cumY,cumYw,sumYw,tlin_rw,tcon_rw --variants "scan_chain=1" --rounds 1 --reps 3
echo "== VALU issue share"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $OUT/prof_valu_r03end -o valu -- python $REPO/tools/microbench.py --reps 5 --cases stencil,metric,cumsum,reduce > $OUT/prof_valu_r03end.log 2>&1
cd $REPO
python tools/valu_util.py $(find $OUT/prof_valu_r03end -name "*.db" | head -1) 2>&1 | tee $S/valu_issue_share.txt | head -30
rm -rf $OUT/prof_stats_* $OUT/prof_fetch_* $OUT/prof_write_* $OUT/prof_valu_*  # raw traces stay on the box
echo "== generated tables"
timeout 900 python tools/roofline_table.py --out $S/roofline 2>&1 | tail -34
timeout 600 python tools/scale_table.py --records 45 --out $S/scale_table 2>&1 | tail -10
echo "== box kind"; bash tools/box_kind_pmc.sh $S/box_kind.txt > /dev/null 2>&1; head -2 $S/box_kind.txt | cut -c1-160
