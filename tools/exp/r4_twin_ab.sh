#!/bin/bash
# round 4: what the compute-only twin's 14 % is made of — twins that keep their gathers (twin_g), their plane loads (twin_p), or
# drop the LDS reads of the rigid matrix as well (twin_t); bench's compute_only_avg_launch_ms against the kernel's avg_launch_ms
out=gpurun_out/$1; mkdir -p $out
cp uw-slam_amd/libuwt_hip.so /tmp/keep.so
for r in 1 2; do
  for v in cur twin_g twin_p twin_t; do
    cp tools/exp/lib_$v.so uw-slam_amd/libuwt_hip.so
    python bench.py --cpu-pairs 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v#$r', d['value'], 'kernel', r['avg_launch_ms'], 'twin', r['valu']['compute_only_avg_launch_ms'], r['valu']['valu_issue_frac'], r['valu']['shader_clock_GHz'])"
  done
done > $out/twin.txt 2>&1
cp /tmp/keep.so uw-slam_amd/libuwt_hip.so
