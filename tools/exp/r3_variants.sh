#!/bin/bash
# per-kernel averages of library variants:  r3_variants.sh "<bench args>" <kernel substring> <libA> <libB> ...
args=$1; pat=$2; shift 2
R=$(pwd); out=$R/gpurun_out/r3var; mkdir -p $out
cp uw-slam_amd/libuwt_hip.so /tmp/lib_saved.so
for v in "$@"; do
  cp tools/exp/lib_$v.so uw-slam_amd/libuwt_hip.so
  (cd /tmp && export TMPDIR=/tmp && cd $R && rocprofv3 --kernel-trace --stats -d $out/$v --output-format csv -- python3 bench.py --cpu-pairs 0 --no-profile $args > $out/$v.json 2> $out/$v.err)
  f=$(find $out/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(python3 -c "import json,sys; d=json.loads(open('$out/$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
  grep "$pat" $f | cut -d, -f1-4 | cut -c1-160
  rm -rf $out/$v
done
cp /tmp/lib_saved.so uw-slam_amd/libuwt_hip.so
