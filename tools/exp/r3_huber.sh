#!/bin/bash
# general path (robust weights / bilinear) at 256 pairs: parity tests first, then the three bench lines
mkdir -p gpurun_out/r3huber
python -m pytest tests/test_robust_bilinear.py tests/test_gpu_production.py -x -q -m gpu 2>&1 | tail -5
for w in huber tukey; do
  python bench.py --pairs 256 --unique 8 --cpu-pairs 8 --weights $w > gpurun_out/r3huber/${w}_p256.json 2> gpurun_out/r3huber/${w}_p256.err
done
python bench.py --pairs 256 --unique 8 --cpu-pairs 8 --bilinear > gpurun_out/r3huber/bilinear_p256.json 2> gpurun_out/r3huber/bilinear_p256.err
python bench.py --pairs 256 --unique 8 --cpu-pairs 8 --bilinear --weights huber > gpurun_out/r3huber/bilinear_huber_p256.json 2> gpurun_out/r3huber/bilinear_huber_p256.err
for f in gpurun_out/r3huber/*.json; do echo "$f: $(python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('parity'))
except Exception as e: print('ERR', e)
PY
)"; done
