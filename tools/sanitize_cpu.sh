#!/bin/bash
# The CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer (there is no GPU sanitizer on this pool, and none is
# attempted): the C oracle that arbitrates every parity claim (oracle/Makefile SAN=1), the benchmarks' input generator, the
# gemm-fold probe CLI and the optional OpenCV / Eigen branches of include/uw_tracker.hpp (tools/Makefile san).
#   tools/sanitize_cpu.sh [log file, default profiles/r06/sanitize_cpu.log]
# Runs: tests/test_oracle.py, the golden generator (into a scratch directory; must reproduce tests/golden/*.npz bit for bit), the
# CPU suite (-m "not gpu"), the generator and the probe — all with the sanitized libraries and the sanitizer runtimes preloaded
# into the interpreter.  Any sanitizer report aborts the process it occurs in (-fno-sanitize-recover, halt_on_error): a clean log
# is a log whose every step says "ok".
set -u
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r06/sanitize_cpu.log}
mkdir -p "$(dirname "$LOG")"
: > "$LOG"
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
say() { echo "$@" | tee -a "$LOG"; }
step() {  # name, command...
  name=$1; shift
  if "$@" >> "$LOG.tmp" 2>&1; then say "ok      $name"; else say "FAILED  $name"; tail -40 "$LOG.tmp" | tee -a "$LOG"; FAIL=1; fi
  grep -E "ERROR: AddressSanitizer|runtime error:|SUMMARY: (Address|UndefinedBehavior)Sanitizer" "$LOG.tmp" | head -20 | tee -a "$LOG"
  grep -E "^[0-9]+ passed|passed in|failed in" "$LOG.tmp" | tail -1 | sed 's/^/        /' | tee -a "$LOG"
  rm -f "$LOG.tmp"
}
FAIL=0
say "# tools/sanitize_cpu.sh — $(gcc --version | head -1); $(date -u +%Y-%m-%dT%H:%MZ)"
step "build oracle (ASan + UBSan)" make -C oracle -B SAN=1
step "build generator, fold probe, optional C++ branches (ASan + UBSan)" make -C tools -B san
# the interpreter itself is not instrumented: leaks of CPython / numpy / torch are not ours to report
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:alloc_dealloc_mismatch=0:detect_odr_violation=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export UWT_ORACLE_SAN=1
PRE="$ASAN:$UBSAN"
SCR=$(mktemp -d /tmp/uwt_san_XXXX)
step "tests/test_oracle.py under the sanitized oracle" env LD_PRELOAD=$PRE python -m pytest tests/test_oracle.py -q -x -p no:cacheprovider
step "golden generator under the sanitized oracle" env LD_PRELOAD=$PRE python tests/golden/make_golden.py $SCR
step "generated goldens equal the committed ones, array by array" python - $SCR <<'PY'
import glob, os, sys
import numpy as np
n = 0
for f in sorted(glob.glob("tests/golden/*.npz")):
    a, b = np.load(f), np.load(os.path.join(sys.argv[1], os.path.basename(f)))
    assert sorted(a.files) == sorted(b.files), f
    for k in a.files:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), (f, k)
        n += 1
print("%d arrays in %d files identical" % (n, len(glob.glob("tests/golden/*.npz"))))
PY
step "the CPU suite (-m 'not gpu') under the sanitized oracle" env LD_PRELOAD=$PRE python -m pytest tests -q -x -m "not gpu" -p no:cacheprovider
step "input generator (sanitized) = input generator (plain), 6 pairs incl. an odd size" env LD_PRELOAD=$PRE python - <<'PY'
import ctypes as C, hashlib
import numpy as np
def gen(path):
    lib = C.CDLL(path)
    lib.uwt_gen_pair.restype = C.c_int
    lib.uwt_gen_pair.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    h = hashlib.sha256()
    for (w, hh, gid) in ((160, 96, 0), (160, 96, 5), (733, 471, 1), (64, 48, 2), (9, 7, 3), (640, 480, 4)):
        ref = np.empty((hh, w), np.uint8); tgt = np.empty((hh, w), np.uint8); dep = np.empty((hh, w), np.uint16); z = C.c_double()
        assert lib.uwt_gen_pair(w, hh, 0.8 * w, 0.8 * w, w / 2 - 0.5, hh / 2 - 0.5, gid, ref.ctypes.data, tgt.ctypes.data, dep.ctypes.data, C.byref(z)) == 0
        for a in (ref, tgt, dep): h.update(a.tobytes())
    return h.hexdigest()
a, b = gen("tools/libuwt_gen_san.so"), gen("tools/libuwt_gen.so")
assert a == b, (a, b)
print("generator digests equal:", a[:16])
PY
step "fold probe CLI (sanitized) on 5 rows" bash -c 'for r in "0x1.fffp-1 0x1.2p-7 -0x1.8p-9" "0x1p0 0x0p0 0x0p0" "-0x1.4p-3 0x1.fp-1 0x1.1p-5" "0x1.8p-2 -0x1.8p-2 0x1.cp-1" "0x1.0p-20 0x1.0p-21 0x1p0"; do ./tools/fold_probe_cli_san $r || exit 1; done'
rm -rf $SCR
if [ $FAIL = 0 ]; then say "ALL STEPS OK: no AddressSanitizer / UBSan report"; else say "SANITIZE RUN FAILED"; fi
exit $FAIL
