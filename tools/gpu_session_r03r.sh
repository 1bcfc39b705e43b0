#!/bin/bash
# Round 3, session R: re-tune launch shapes after the band-height change (levels / rows per wave-task of the metric
# kernels), sub-band widths of the chained weighted reduction
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03r
mkdir -p $S
export TMPDIR=/tmp
echo "== parity (after the X scan revert)"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_f32.py -x -q 2>&1 | tail -2 | tee $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases dY,iYmw --variants "met_zk1=2,met_seg1=2;met_zk1=4,met_seg1=2;met_zk1=2,met_seg1=4;met_zk1=1,met_seg1=2;met_zk=4,met_seg=4;met_zk=2,met_seg=4;met_zk=4,met_seg=2;met_zk=2,met_seg=2" --rounds 4 --reps 5 2>&1 | grep '^{' | tee $S/ab_k2s_shapes.jsonl | cut -c1-150
timeout 300 python tools/ab_tunables.py --cases dX,iXmw --variants "contig_rw=2,contig_rw_mi=8;contig_rw=4,contig_rw_mi=8;contig_rw=2,contig_rw_mi=4;contig_rw=1,contig_rw_mi=4;contig_rw=2,contig_rw_mi=2" --rounds 4 --reps 5 2>&1 | grep '^{' | tee $S/ab_k1r_shapes.jsonl | cut -c1-150
timeout 300 python tools/ab_tunables.py --cases sumYw,cumYw --variants "scan_chain_w=1;scan_chain_w=102;scan_chain_w=104;scan_chain_w=108;reduce_zl=1;reduce_zl=4" --rounds 4 --reps 5 2>&1 | grep '^{' | tee $S/ab_chain_w.jsonl | cut -c1-150
