#!/bin/bash
# Round 3, session BF: lean transforms: rows leave as 16-B stores of lane pairs (transform_lean bit 2), with `sc1 nt` (bit 3)
S=$PWD/gpurun_out/r03bf
mkdir -p $S
export TMPDIR=/tmp
for v in 7 15; do echo "== parity transform_lean=$v"; XG_TRANSFORM_LEAN=$v timeout 900 python -m pytest tests/test_transform.py -x -q -m gpu 2>&1 | tail -1 | tee -a $S/pytest.log; done
XG_TRANSFORM_LEAN=15 timeout 300 python tools/fuzz_transform_variants.py 2>&1 | tail -1 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases tlin_rw,tlin_sm,tcon_rw,tcon_sm --variants "transform_lean=3;transform_lean=7;transform_lean=15" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_row_store.jsonl | cut -c1-150
