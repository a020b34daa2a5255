#!/bin/bash
# round-3 starting point on today's box: default line, identity / huber at 256 pairs, and the identity path at the finest slicing
# (2 groups per thread in every launch: what a per-block overhead costs)
mkdir -p gpurun_out/r3base
python bench.py > gpurun_out/r3base/default.json 2> gpurun_out/r3base/default.err
python bench.py --pairs 256 --unique 8 --cpu-pairs 0 > gpurun_out/r3base/ident_p256.json 2>&1
python bench.py --pairs 256 --unique 8 --cpu-pairs 0 --weights huber > gpurun_out/r3base/huber_p256.json 2>&1
UWT_TARGET_BLOCKS=100000000 python bench.py --pairs 256 --unique 8 --cpu-pairs 0 --no-profile > gpurun_out/r3base/ident_p256_finest.json 2>&1
UWT_TARGET_BLOCKS=100000000 UWT_GROUPS_PER_THREAD=4 python bench.py --pairs 256 --unique 8 --cpu-pairs 0 --no-profile > gpurun_out/r3base/ident_p256_g4.json 2>&1
UWT_TARGET_BLOCKS=100000000 UWT_GROUPS_PER_THREAD=8 python bench.py --pairs 256 --unique 8 --cpu-pairs 0 --no-profile > gpurun_out/r3base/ident_p256_g8.json 2>&1
python bench.py --reference-schedule --cpu-pairs 0 --no-profile > gpurun_out/r3base/refsched.json 2>&1
for f in gpurun_out/r3base/*.json; do echo "$f: $(python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))
except Exception as e: print('ERR', e)
PY
)"; done
