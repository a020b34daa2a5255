"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol include/uwt.h declares,
and refuses to run without a gfx950 device (no CPU fallback).  No compute calls here."""
import importlib
import os
import re
import sys

ARITH_INDEPENDENT = True   # nothing here depends on the arithmetic set (tests/conftest.py): run once

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    importlib.import_module("uw-slam_amd").build_native()
    return importlib.import_module("uw-slam_amd.capi")


def header_symbols():
    src = open(os.path.join(ROOT, "include", "uwt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uwt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(capi):
    lib = capi.lib()
    declared = header_symbols()
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(capi.SYMBOLS) == declared
    assert lib.uwt_abi_version() == 4   # 3: uwt_tuning; 4: any frame size (uwt_level::img_w / img_h / pitch, uwt_resize_half_*)


def test_source_id_is_the_hash_of_sources_and_flags(capi):
    """uwt_source_id(): what bench.py compares with the id stamped on profiles/rNN/k_residual_facts.json.  The Makefile hashes
    its source files (every translation unit and header) followed by its flags line; recomputed here from the same inputs."""
    import hashlib
    import subprocess
    sid = capi.source_id()
    assert re.fullmatch(r"[0-9a-f]{64}", sid), sid
    csrc = os.path.join(ROOT, "uw-slam_amd", "csrc")
    def make_var(name):
        return subprocess.run(["make", "-s", "-C", csrc, "--eval", "print-var: ; @echo '$(%s)'" % name, "print-var"],
                              capture_output=True, text=True, check=True).stdout
    flags, sources = make_var("CXXFLAGS"), make_var("SOURCES").split()
    assert "uwt_capi.hip" in sources and "uwt_kernels.h" in sources and "../../include/uwt.h" in sources
    h = hashlib.sha256()
    for f in sources:
        h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(flags.encode())
    assert h.hexdigest() == sid


def test_struct_layouts_match_header(capi):
    import ctypes as C
    assert C.sizeof(capi.Params) == 26 * 4
    assert C.sizeof(capi.Level) == 11 * 4
    assert C.sizeof(capi.Stats) == 16
    assert C.sizeof(capi.Accum) == 21 * 8 + 6 * 8 + 8 + 8
    assert C.sizeof(capi.Tuning) == 2 * 4 + 2 * 8 + 12 * 4 + 4 * 4


def test_library_reads_no_environment_and_carries_no_experiment_switches():
    """The production surface: launch shapes are chosen through uwt_tuning, results through uwt_params — no getenv anywhere in
    the library's sources, no UWT_EXP_* experiment branch in the kernels."""
    csrc = os.path.join(ROOT, "uw-slam_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            txt = open(os.path.join(csrc, f)).read()
            assert "getenv" not in txt, f
            assert "UWT_EXP_" not in txt, f


def test_default_params_are_the_reference_constants(capi):
    p = capi.default_params(640, 480, 525.0, 525.0, 319.5, 239.5)
    # src/Tracker.cpp:364-372, 393, 559, 1261; src/Options.cpp:26
    assert (p.n_levels, p.first_level, p.last_level, p.max_iters) == (5, 4, 1, 50)
    assert p.epsilon == pytest.approx(1e-3) and p.gain == 50.0 and p.initial_error == 50000.0
    assert p.z_factor == 1.0 and p.angle_factor == 1.0 and p.depth_scale == pytest.approx(2e-4)
    assert p.early_exit == 1 and p.handoff_scale_t == 0
    assert capi.lib().uwt_status_string(2) == b"no valid points"


def test_no_gpu_means_loud_failure_not_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = capi.default_params(64, 48, 64.0, 64.0, 31.5, 23.5, n_levels=3, first_level=2, last_level=0)
    with pytest.raises(capi.UwtError) as e:
        capi.Context(p)
    assert e.value.status == capi.ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under uw-slam_amd/ or include/ may reference it."""
    bad = []
    for base in ("uw-slam_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"\boracle\b", txt) and f not in ("__init__.py",):
                        for line in txt.splitlines():
                            if re.search(r"import .*oracle|from oracle|uwt_oracle|libuwt_oracle|oracle/", line):
                                bad.append((f, line.strip()))
    assert not bad, bad


def test_hand_written_loads_of_the_typed_kernels_are_released_before_use():
    """The typed instantiations of k_residual (uwt_kernels.h: load_group_typed) issue every vector-memory operation of their loop
    from asm statements, which the compiler's s_waitcnt bookkeeping does not see: between such a load and the written-out wait
    that releases it, no instruction of the SHIPPED assembly may touch the load's destination registers.  Compiles the
    dispatcher unit to assembly (the library's own `make asm` recipe) and scans every typed instantiation
    (tools/check_asm_loads.py)."""
    import subprocess
    csrc = os.path.join(ROOT, "uw-slam_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asm", "UNIT=uwt_launch_residual"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = os.path.join(csrc, "uwt_launch_residual.gfx950.s")
    names = sorted(set(re.findall(r"^(_ZN3uwt10k_residualI\w+?ELi[23]EEEvNS_12ResidualArgsE):", open(asm).read(), re.M)))
    assert len(names) >= 8, names            # production + EUROC forms x depth x streamed, both arithmetic sets
    for n in names:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_loads.py"), asm, n], capture_output=True, text=True)
        assert out.returncode == 0 and " 0 violations" in out.stdout, (n, out.stdout[-1500:])
        assert int(out.stdout.split()[0]) >= 16, out.stdout          # the loop's loads were found (two register sets)
