#!/bin/bash
# ONE parametrised GPU-box session (replaces the per-session scripts of rounds 1-3, which live in the git history):
#
#     bash tools/gpu_session.sh <name> <steps-file>        # on the GPU box, from the repo root
#     gpurun --timeout 1500 -- 'bash tools/gpu_session.sh r04a tools/sessions/r04a.steps'
#
# <steps-file>: one step per line (# comments), executed in order; everything a step writes lands in gpurun_out/<name>/,
# from where the files worth keeping are copied to profiles/<name>_* by hand.  Steps:
#     pytest [pytest args]        the GPU suite (default: tests -m gpu -q)            -> pytest_gpu.log
#     smoke                       __graft_entry__.smoke()                              -> smoke.log
#     bench [bench.py args]       the driver's bench line                             -> bench_1gpu.json
#     bench_rccl                  the same through RCCL with one rank                  -> bench_rccl_ws1.json
#     configs                     tools/bench_configs.py 2,3 / 4,5 (45 records) / 5x,f1 -> configs_*.jsonl
#     tables                      tools/microbench.py f64 + f32                        -> microbench_*.jsonl
#     survey [args]               tools/survey.py                                      -> survey.jsonl
#     survey_trace [args]         tools/survey.py --trace (kernel time next to wall)   -> survey_trace.jsonl
#     prof <tag> <command...>     rocprofv3 --kernel-trace --stats, then separate --pmc FETCH_SIZE / WRITE_SIZE passes of
#                                 <command> (never combined with other trace domains) -> rocprof_summary_<tag>.txt
#     valu                        SQ_INSTS_VALU / SALU / WAVES over the kernel table    -> valu_issue_share.txt
#     roofline [args]             tools/roofline_table.py                              -> roofline.md / .jsonl
#     scale [args]                tools/scale_table.py --records 45                    -> scale_table.md / .jsonl / _topo.txt
#     run <label> <secs> <cmd..>  any command under `timeout <secs>`                    -> <label>.log
NAME=${1:?session name}
STEPS=${2:?steps file}
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/$NAME
mkdir -p $S
export TMPDIR=/tmp
prof() {
  local tag=$1; shift
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$tag -o stats -- "$@" > $OUT/prof_stats_$tag.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch_$tag -o fetch -- "$@" > $OUT/prof_fetch_$tag.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write_$tag -o write -- "$@" > $OUT/prof_write_$tag.log 2>&1
  cd $REPO
  python tools/summarize_prof.py $OUT $tag > $S/rocprof_summary_$tag.txt 2>&1
  cp $OUT/pmc_traffic_$tag.json $S/ 2>/dev/null
  rm -rf $OUT/prof_stats_$tag $OUT/prof_fetch_$tag $OUT/prof_write_$tag  # raw traces stay on the box
  echo "-- $tag"; head -8 $S/rocprof_summary_$tag.txt
}
while IFS= read -r line || [ -n "$line" ]; do
  line="${line%%#*}"
  set -- $line
  [ $# -eq 0 ] && continue
  step=$1; shift
  echo "== $step $*"
  case $step in
    pytest) timeout 900 python -m pytest ${@:-tests -m gpu -q} 2>&1 | tail -15 | tee $S/pytest_gpu.log ;;
    smoke) timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $S/smoke.log ;;
    bench) timeout 400 python bench.py "$@" 2>&1 | tail -1 | tee $S/bench_1gpu.json ;;
    bench_rccl) XG_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee $S/bench_rccl_ws1.json ;;
    configs)
      timeout 300 python tools/bench_configs.py --configs 2,3 2>&1 | grep '^{' | tee $S/configs_2_3.jsonl | cut -c1-220
      timeout 400 python tools/bench_configs.py --gpus 1 --configs 4,5 --records 45 2>&1 | grep '^{' | tee $S/configs_4_5_sharded_1rank.jsonl | cut -c1-260
      timeout 300 python tools/bench_configs.py --configs 5x,f1 2>&1 | grep '^{' > $S/configs_5x_f1.jsonl ;;
    tables)
      timeout 200 python tools/microbench.py --reps 9 --cases copy,stencil,metric,cumsum,reduce,vort,generic 2>/dev/null | grep '^{' > $S/microbench_all_kernels.jsonl
      timeout 200 python tools/microbench.py --reps 9 --dtype f32 --cases stencil,metric,cumsum,reduce,vort 2>/dev/null | grep '^{' > $S/microbench_f32.jsonl ;;
    survey) timeout 300 python tools/survey.py --reps 7 "$@" 2>&1 | grep '^{' | tee $S/survey.jsonl | cut -c1-200 ;;
    survey_trace) timeout 600 python tools/survey.py --reps 7 --trace "$@" 2>&1 | grep '^{' | tee $S/survey_trace.jsonl | cut -c1-260 ;;
    prof) prof "$@" ;;
    valu)
      cd /tmp
      timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $OUT/prof_valu_$NAME -o valu -- python $REPO/tools/microbench.py --reps 5 --cases stencil,metric,cumsum,reduce > $OUT/prof_valu_$NAME.log 2>&1
      cd $REPO
      python tools/valu_util.py $(find $OUT/prof_valu_$NAME -name "*.db" | head -1) 2>&1 | tee $S/valu_issue_share.txt | head -30
      rm -rf $OUT/prof_valu_$NAME ;;
    roofline) timeout 1200 python tools/roofline_table.py --out $S/roofline "$@" 2>&1 | tail -40 ;;
    scale) timeout 600 python tools/scale_table.py --allow-skips --records 45 --out $S/scale_table "$@" 2>&1 | tail -10 ;;
    run) label=$1; secs=$2; shift 2; timeout $secs "$@" 2>&1 | tee $S/$label.log | tail -40 ;;
    *) echo "unknown step: $step" ;;
  esac
done < "$STEPS"
