#!/bin/bash
# round 4: the 4-wave k_coarse / bilinear+Huber variants and the build without machine LICM (lib_w4) against the commit before (lib_base)
out=gpurun_out/$1; mkdir -p $out
cp tools/exp/lib_w4.so /tmp/keep_w4.so
{
bash tools/exp/ab_args.sh 3 "" base w4
bash tools/exp/ab_args.sh 2 "--weights huber --pairs 256 --unique 8" base w4
bash tools/exp/ab_args.sh 2 "--bilinear --weights huber --pairs 256 --unique 8" base w4
bash tools/exp/ab_args.sh 2 "--weights huber --unique 8" base w4
bash tools/exp/ab_args.sh 2 "--reference-schedule" base w4
} > $out/ab.txt 2>&1
cp /tmp/keep_w4.so uw-slam_amd/libuwt_hip.so
python bench.py > $out/bench.json 2> $out/bench.err
