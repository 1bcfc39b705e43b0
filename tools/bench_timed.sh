#!/bin/bash
# `python bench.py "$@"` under `time`, the way the driver runs it: JSON line -> $OUT_JSON (default gpurun_out/bench_timed.json),
# stderr -> next to it; prints the wall time, the line's size and the line (gpu_session.sh `run` steps cannot quote)
OUT_JSON=${OUT_JSON:-gpurun_out/bench_timed.json}
mkdir -p "$(dirname "$OUT_JSON")"
START=$(date +%s.%N)
python bench.py "$@" > "$OUT_JSON" 2> "${OUT_JSON%.json}_stderr.log"
RC=$?
END=$(date +%s.%N)
echo "rc=$RC wall_s=$(python -c "print(round($END-$START,1))") bytes=$(wc -c < "$OUT_JSON")"
tail -c 9000 "$OUT_JSON"
tail -5 "${OUT_JSON%.json}_stderr.log"
