#!/bin/bash
# Round 3, session AN: the weighted march with outer indices fastest (weights from the L2) against K4L
S=$PWD/gpurun_out/r03an
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; XG_REDUCE_LDSW=0 XG_SCAN_CHAIN=0 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases sumYw,avgYw --variants "reduce_ldsw=2;reduce_ldsw=0,scan_chain=0,march_ofast=0;reduce_ldsw=0,scan_chain=0,march_ofast=1" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_march_ofast.jsonl | cut -c1-190
timeout 300 python tools/pmc_ab.py --cases sumYw --variants "reduce_ldsw=0,scan_chain=0,march_ofast=0;reduce_ldsw=0,scan_chain=0,march_ofast=1" --pmc "FETCH_SIZE" 2>&1 | tee $S/pmc_march_ofast.jsonl | cut -c1-330
