#!/bin/bash
# The round's whole verification on one box, results under gpurun_out/<name>/ :  tools/verify_all.sh <name>
#   the GPU test suite, the default bench line (with the quoted counter facts when profiles/<round>/k_residual_facts.json matches the library),
#   the parity survey, and every fuzz tool of tools/exp (odd sizes, ROI-like sizes, one long-lived context, point tables, extreme poses, auxiliary entry points,
#   the asynchronous pipeline, frame-by-frame sequences, many threads, every upload form).
out=gpurun_out/${1:-verify}; mkdir -p $out
python -c "import ctypes; h = ctypes.CDLL('uw-slam_amd/libuwt_hip.so'); h.uwt_source_id.restype = ctypes.c_char_p; print(h.uwt_source_id().decode())" > $out/library_source_id.txt
python -m pytest tests -m gpu -q 2>&1 | tail -2 > $out/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $out/bench_default_p1024_final.json 2> $out/bench_final.err
bash tools/parity_survey.sh ${1:-verify}/parity_survey > /dev/null 2>&1
bash tools/exp/odd_shapes.sh ${1:-verify}/odd_shapes > /dev/null 2>&1
bash tools/exp/roi_shapes.sh ${1:-verify}/roi_shapes > /dev/null 2>&1     # frame sizes the resize chain does not divide (round 6)
F="grep -v amdgpu.ids"
for s in 2 3 4; do python tools/exp/points_fuzz.py 1500 $s 2>&1 | $F; done > $out/points_fuzz.txt
(for s in 2 3 4 5; do python tools/exp/stateful_fuzz.py 1500 $s 2>&1 | $F | cut -c1-1500; done; FUZZ_DEEP=1 python tools/exp/stateful_fuzz.py 800 11 2>&1 | $F | cut -c1-1500) > $out/stateful_fuzz.txt
for s in 2 3 4; do python tools/exp/stage_fuzz.py 3000 $s 2>&1 | $F | cut -c1-900; done > $out/stage_fuzz.txt
for s in 2 3; do python tools/exp/aux_fuzz.py 3000 $s 2>&1 | $F | cut -c1-1500; done > $out/aux_fuzz.txt
for s in 2 3 4; do python tools/exp/stream_fuzz.py 4000 $s 2>&1 | $F | cut -c1-600; done > $out/stream_fuzz.txt
for s in 2 3 4; do python tools/exp/sequence_fuzz.py 400 $s 2>&1 | $F | cut -c1-700; done > $out/sequence_fuzz.txt
for a in "6 3000 2" "12 2000 3" "8 3000 4"; do timeout 500 python tools/exp/thread_fuzz.py $a 2>&1 | $F | cut -c1-700; done > $out/thread_fuzz.txt
for s in 2 3 4; do python tools/exp/upload_fuzz.py 400 $s 2>&1 | $F | cut -c1-400; done > $out/upload_fuzz.txt
cat $out/pytest_gpu.txt; tail -c 250 $out/bench_default_p1024_final.json; echo
for f in points_fuzz stateful_fuzz stage_fuzz aux_fuzz stream_fuzz sequence_fuzz thread_fuzz upload_fuzz; do echo "== $f"; grep -i "differ\|alignments:" $out/$f.txt | grep -v "so far" | tail -6 | cut -c1-300; done
grep -c "bit-identical" $out/parity_survey.txt
tail -1 $out/odd_shapes.txt; tail -1 $out/roi_shapes.txt
