#!/bin/bash
# round 4: the general path's one-unit-at-a-time form (bilinear sampler, per-pixel weights) — lib_cur — against the commit before it (lib_base)
out=gpurun_out/$1; mkdir -p $out
cp tools/exp/lib_cur.so /tmp/keep_cur.so
{
bash tools/exp/ab_args.sh 3 "--bilinear --weights huber --pairs 256 --unique 8" base cur
bash tools/exp/ab_args.sh 2 "--bilinear --pairs 256 --unique 8" base cur
bash tools/exp/ab_args.sh 2 "--arith legacy --bilinear --weights huber --pairs 256 --unique 8" base cur
bash tools/exp/ab_args.sh 2 "--weights huber --pairs 256 --unique 8" base cur
} > $out/ab.txt 2>&1
cp /tmp/keep_cur.so uw-slam_amd/libuwt_hip.so
