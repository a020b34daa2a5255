"""The header-only C++ mirror (include/uw_tracker.hpp): compiles and links against libuwt_hip.so on CPU; on the GPU
it runs System::Tracking()'s call sequence and must reproduce the oracle's pose."""
import importlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "shim_sequence")


def build_exe():
    importlib.import_module("uw-slam_amd").build_native()
    libdir = os.path.join(ROOT, "uw-slam_amd")
    src = os.path.join(ROOT, "tests", "cpp", "shim_sequence.cpp")
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(src), os.path.getmtime(os.path.join(libdir, "libuwt_hip.so")),
                                                          os.path.getmtime(os.path.join(ROOT, "include", "uw_tracker.hpp"))):
        return EXE
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                           "-L", libdir, "-luwt_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_shim_compiles_and_links():
    assert os.path.exists(build_exe())


@pytest.mark.gpu
def test_shim_sequence_matches_oracle(O, synth, tmp_path):
    exe = EXE if os.path.exists(EXE) else build_exe()
    w, h = 160, 96
    f = 525.0 * w / 640.0
    ref, tgt, _, _, _ = synth.render_pair(w, h, f, f, w / 2 - 0.5, h / 2 - 0.5, seed=77)
    raw = tmp_path / "pair.raw"
    raw.write_bytes(ref.tobytes() + tgt.tobytes())
    out = subprocess.run([exe, str(raw), str(w), str(h)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    vals = lines[0].split()
    pose = np.array([float(v) for v in vals[:7]], np.float32)
    st, pose_cpu, tr = O.align_pair(O.default_params(w, h, f, f, w / 2 - 0.5, h / 2 - 0.5), ref, tgt, want_trace=True)
    assert st == 0 and int(vals[7]) == len(tr)
    assert np.array_equal(pose, pose_cpu)
    ft = lines[1].split()
    assert ft[0] == "FEATURES"
    kp = np.array([[8.0 + (k * 37) % (w - 16), 8.0 + (k * 23) % (h - 16)] for k in range(40)], np.float32)
    pts, n = O.patch_points(kp, None, w, h)
    feat = dict(n_levels=5, first_level=0, last_level=0, max_iters=10, early_exit=1, gain=1.0, z_factor=0.002, handoff_scale_t=1)
    so, pose_f, tr_f = O.align_pair_points(O.default_params(w, h, f, f, w / 2 - 0.5, h / 2 - 0.5, **feat), ref, tgt, {0: pts},
                                           want_trace=True)
    assert so == 0 and int(ft[9]) == n and int(ft[8]) == len(tr_f)
    assert np.array_equal(np.array([float(v) for v in ft[1:8]], np.float32), pose_f)
    ls = lines[2].split()
    assert ls[0] == "LS" and float(ls[1]) == 1.0 and float(ls[2]) == -3.0 and float(ls[3]) == 2.0 and int(ls[4]) == 1


BENCH = os.path.join(ROOT, "tools", "uwt_bench")


def build_bench():
    importlib.import_module("uw-slam_amd").build_native()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
    return BENCH


def test_native_bench_compiles_and_links():
    assert os.path.exists(build_bench())


@pytest.mark.gpu
def test_native_bench_runs_without_python_in_the_loop():
    """tools/uwt_bench: C++ generator + the C ABI only.  Poses finite, tiled copies of a pair bit-identical."""
    import json
    exe = BENCH if os.path.exists(BENCH) else build_bench()
    out = subprocess.run([exe, "--pairs", "24", "--unique", "5", "--width", "160", "--height", "96", "--levels", "3", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["poses_finite"] and d["tiled_pairs_identical"] and d["value"] > 0 and 0 < d["max_translation_m"] < 0.1
    out = subprocess.run([exe, "--pairs", "4", "--unique", "4", "--width", "160", "--height", "96", "--reference-schedule", "--no-depth",
                          "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
