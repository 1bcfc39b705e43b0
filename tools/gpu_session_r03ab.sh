#!/bin/bash
# Round 3, session AB: flat K1, neighbour by DPP against the unchanged load path (both compiled in, chosen per launch)
S=$PWD/gpurun_out/r03ab
mkdir -p $S
export TMPDIR=/tmp
timeout 300 python tools/ab_tunables.py --cases diffX,diffY,dX --variants "nb_dpp=0;nb_dpp=1" --rounds 8 --reps 7 2>&1 | grep '^{' | tee $S/ab_k1dpp.jsonl | cut -c1-150
