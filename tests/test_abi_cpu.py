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
    that releases it, no instruction may touch the load's destination registers, on any path (branch targets and loop back-edges
    followed), and no such load may reach s_endpgm unwaited.  Checked on the code objects INSIDE THE SHIPPED libuwt_hip.so
    (llvm-objdump --offloading + -d; hipcc's register allocation differs from build to build, so a fresh compile proves nothing
    about the binary that ships) and, for the tool's other input format, on a fresh listing of the dispatcher unit."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "check_asm_loads.py")
    lib = os.path.join(ROOT, "uw-slam_amd", "libuwt_hip.so")
    out = subprocess.run([sys.executable, tool, "--shipped", lib, "--all-typed"], capture_output=True, text=True)
    assert out.returncode == 0 and " 0 violations" in out.stdout, out.stdout[-3000:] + out.stderr[-1000:]
    m = re.search(r"(\d+) vector loads checked in (\d+) kernels", out.stdout)
    # production + EUROC (fx != fy) forms x depth x streamed x whole / ragged rows, both arithmetic sets; two register sets of
    # (1 + 2 or 3) plane loads + 4 gathers per kernel and the first group's request
    assert int(m.group(2)) >= 32 and int(m.group(1)) >= 16 * int(m.group(2)), out.stdout
    csrc = os.path.join(ROOT, "uw-slam_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asm", "UNIT=uwt_launch_residual"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([sys.executable, tool, os.path.join(csrc, "uwt_launch_residual.gfx950.s"), "--all-typed"], capture_output=True, text=True)
    assert out.returncode == 0 and " 0 violations" in out.stdout, out.stdout[-3000:]


def test_the_load_checker_catches_what_it_is_for(tmp_path):
    """tools/check_asm_loads.py on hand-made listings: a use before the wait, a younger store whose data register is the load's
    destination, a load that reaches s_endpgm unwaited, a use that is only reachable through the loop's back-edge — and the
    packed-f32 broadcast that names a register pair but reads its low half only."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "check_asm_loads.py")

    def run(body):
        f = tmp_path / "k.s"
        f.write_text("_ZN3uwt6k_testEv:\n" + "".join("\t%s\n" % l if not l.endswith(":") else l + "\n" for l in body) + "\ts_endpgm\n")
        out = subprocess.run([sys.executable, tool, str(f), "k_test"], capture_output=True, text=True)
        return out.returncode, out.stdout

    ok = ["tbuffer_load_format_xyzw v[4:7], v1, s[8:11], 0 offen", "v_add_f32 v2, v3, v3", "s_waitcnt vmcnt(0)", "v_add_f32 v2, v4, v4"]
    assert run(ok)[0] == 0
    assert run(["tbuffer_load_format_xyzw v[4:7], v1, s[8:11], 0 offen", "v_add_f32 v2, v5, v3", "s_waitcnt vmcnt(0)"])[0] == 1
    assert run(["tbuffer_load_format_xyzw v[4:7], v1, s[8:11], 0 offen", "global_store_dword v1, v6, s[2:3]", "s_waitcnt vmcnt(0)"])[0] == 1
    rc, txt = run(["tbuffer_load_format_xyzw v[4:7], v1, s[8:11], 0 offen", "v_add_f32 v2, v3, v3"])
    assert rc == 1 and "s_endpgm" in txt
    loop = [".LBB0_1:", "v_add_f32 v2, v4, v4", "s_waitcnt vmcnt(0)", "tbuffer_load_format_xyzw v[4:7], v1, s[8:11], 0 offen",
            "s_cbranch_scc1 .LBB0_1", "s_waitcnt vmcnt(0)"]
    assert run(loop)[0] == 1                                   # the use at the loop's head comes before the head's wait
    assert run([".LBB0_1:", "s_waitcnt vmcnt(0)", "v_add_f32 v2, v4, v4", "tbuffer_load_format_xyzw v[4:7], v1, s[8:11], 0 offen",
                "s_cbranch_scc1 .LBB0_1", "s_waitcnt vmcnt(0)"])[0] == 0
    # a wait that leaves N younger operations outstanding releases the load only once N younger ones were issued
    assert run(["global_load_ubyte v9, v1, s[2:3]", "tbuffer_load_format_xyzw v[4:7], v1, s[8:11], 0 offen", "s_waitcnt vmcnt(1)", "v_add_u32 v2, v9, v9",
                "s_waitcnt vmcnt(0)"])[0] == 0
    assert run(["global_load_ubyte v9, v1, s[2:3]", "s_waitcnt vmcnt(1)", "v_add_u32 v2, v9, v9", "s_waitcnt vmcnt(0)"])[0] == 1
    bcast = ["global_load_ubyte v45, v1, s[2:3]", "v_pk_add_f32 v[92:93], v[44:45], s[58:59] op_sel_hi:[0,1]", "s_waitcnt vmcnt(0)"]
    assert run(bcast)[0] == 0                                  # both lanes read v44
    assert run([bcast[0], "v_pk_add_f32 v[92:93], v[44:45], s[58:59]", bcast[2]])[0] == 1
