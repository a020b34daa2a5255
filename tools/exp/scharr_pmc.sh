#!/bin/bash
R=$(pwd); out=$R/gpurun_out/scharr_pmc; rm -rf $out; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scharr or gradient or random_shapes" 2>&1 | tail -2
for i in 1 2 3; do python tools/exp/scharr_timing.py 2>/dev/null | tail -1; done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c --output-format csv -- python3 tools/exp/scharr_timing.py > $out/pmc_$c.log 2>&1)
done
cd $R
python3 tools/pmc_summary.py $out/pmc_FETCH_SIZE $out/fetch.csv; python3 tools/pmc_summary.py $out/pmc_WRITE_SIZE $out/write.csv
grep -i scharr $out/fetch.csv $out/write.csv
