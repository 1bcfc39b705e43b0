#!/bin/bash
# Round 3, session AY: on-demand robustness runs on the last build (extra fuzz seeds; transform output-staging / lean variants)
S=$PWD/gpurun_out/r03ay
mkdir -p $S
export TMPDIR=/tmp
timeout 600 python tools/fuzz_extra_seeds.py 2>&1 | tail -2 | tee $S/fuzz_extra_seeds.txt
timeout 600 python tools/fuzz_transform_variants.py 2>&1 | tail -2 | tee $S/fuzz_transform_variants.txt
