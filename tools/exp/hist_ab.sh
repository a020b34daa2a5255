#!/bin/bash
# two library variants (tools/exp/lib_<name>.so) on the robust-weight path: value and per-level evaluation times
python -m pytest tests/test_robust_bilinear.py tests/test_gpu_production.py -x -q -m gpu 2>&1 | tail -1
for r in 1 2 3; do for v in "$@"; do cp tools/exp/lib_$v.so uw-slam_amd/libuwt_hip.so
python bench.py --pairs 256 --unique 8 --cpu-pairs 8 --weights huber 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v huber256', d['value'], [round(l['avg_ms_per_evaluation']*1e3,1) for l in d['roofline']['per_level']], d['parity']['bit_identical_poses'])"
done; done
for v in "$@"; do cp tools/exp/lib_$v.so uw-slam_amd/libuwt_hip.so
python bench.py --cpu-pairs 8 --unique 8 --weights huber 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v huber1024', d['value'], d['roofline']['frac'], d['parity']['bit_identical_poses'])"
python bench.py --pairs 256 --unique 8 --cpu-pairs 8 --weights tukey 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v tukey256', d['value'], d['roofline']['frac'], d['parity']['bit_identical_poses'])"
done
