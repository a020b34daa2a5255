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


def test_optional_opencv_and_eigen_branches_pass_a_syntax_check():
    """UW_WITH_OPENCV / UW_WITH_EIGEN of include/uw_tracker.hpp cannot be compiled against the real libraries here (the image
    has neither); tests/cpp/stubs/ declares just the members the header touches, so that the cv::Mat overloads, the Eigen
    typedefs and the reference's DSO-way block at least go through the compiler's parser and overload resolution."""
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "tests", "cpp", "stubs"), os.path.join(ROOT, "tests", "cpp", "shim_optional_branches.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_shim_sequence_matches_oracle(O, synth, tmp_path, arith):
    exe = EXE if os.path.exists(EXE) else build_exe()
    w, h = 160, 96
    f = 525.0 * w / 640.0
    ref, tgt, _, _, _ = synth.render_pair(w, h, f, f, w / 2 - 0.5, h / 2 - 0.5, seed=77)
    raw = tmp_path / "pair.raw"
    raw.write_bytes(ref.tobytes() + tgt.tobytes())
    out = subprocess.run([exe, str(raw), str(w), str(h), arith], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = {ln.split()[0]: ln.split()[1:] for ln in out.stdout.strip().splitlines()}
    vals = lines["POSE"]
    pose = np.array([float(v) for v in vals[:7]], np.float32)
    intr = (f, f, w / 2 - 0.5, h / 2 - 0.5)
    st, pose_cpu, tr = O.align_pair(O.default_params(w, h, *intr), ref, tgt, want_trace=True)
    assert st == 0 and int(vals[7]) == len(tr)
    assert np.array_equal(pose, pose_cpu)
    ft = lines["FEATURES"]
    kp = np.array([[8.0 + (k * 37) % (w - 16), 8.0 + (k * 23) % (h - 16)] for k in range(40)], np.float32)
    pts, n = O.patch_points(kp, None, w, h)
    feat = dict(n_levels=5, first_level=0, last_level=0, max_iters=10, early_exit=1, gain=1.0, z_factor=0.002, handoff_scale_t=1)
    so, pose_f, tr_f = O.align_pair_points(O.default_params(w, h, *intr, **feat), ref, tgt, {0: pts}, want_trace=True)
    assert so == 0 and int(lines["NPATCH"][0]) == n and int(ft[7]) == len(tr_f)
    assert np.array_equal(np.array([float(v) for v in ft[:7]], np.float32), pose_f)
    ls = lines["LS"]
    assert float(ls[0]) == 1.0 and float(ls[1]) == -3.0 and float(ls[2]) == 2.0 and int(ls[3]) == 1
    # the reference's commented "DSO-way" block (src/Tracker.cpp:537-550) compiled as written: LS ls; update(Mat61f, ..);
    # finish(); A.ldlt().solve(-b) — against the oracle's LS on the same rows
    Jm = np.array([[np.float32(0.5) * np.float32(((i + 1) * (k + 2) * 7) % 13) - np.float32(2.0) + (np.float32(4.0) if i % 6 == k else np.float32(0.0))
                    for k in range(6)] for i in range(12)], np.float32)
    rv = np.array([0.75 - 1.5 * (i % 5) for i in range(12)], np.float32)
    wv = np.array([1.0 + 0.25 * (i % 5) for i in range(12)], np.float32)
    o = O.ls_new()
    for i in range(12):
        O.ls_update(o, Jm[i], rv[i], wv[i])
    A, bvec, err, cnt = O.ls_finish(o, divide=True)
    dso = lines["DSO"]
    assert int(dso[0]) == cnt == 12 and int(dso[3]) == 1
    assert np.allclose([float(dso[1]), float(dso[2])], [A[2, 3], bvec[4]], rtol=1e-5, atol=1e-5)
    # Tracker::Mat2SE3 and Tracker::AddPatchPointsFeatures
    m2 = np.array([float(v) for v in lines["MAT2SE3"]], np.float32)
    assert np.array_equal(m2[:4], O.se3_exp(np.array([0, 0, 0, 0.02, -0.01, 0.03], np.float32))[:4])
    assert np.array_equal(m2[4:], np.array([0.5, -0.25, 0.125], np.float32))
    tab = np.array([[3.4, 2.6, 0.7, 1.0], [0.2, 0.4, 1.5, 1.0], [40.5, 20.5, 1.1, 1.0]], np.float32)
    want_pts, n_pts = O.add_patch_points(tab, w // 2, h // 2)
    ap = lines["ADDPATCH"]
    assert int(ap[0]) == n_pts and float(ap[1]) == pytest.approx(float(want_pts.astype(np.float64).sum()), rel=1e-6)
    # LS::updateSSE twice + finish(): the oracle's 4-wide LS on the same operands (count quirk: 6 per call)
    o = O.ls_new()
    for call in range(2):
        J = np.zeros((6, 4), np.float32); res = np.zeros(4, np.float32); wgt = np.zeros(4, np.float32)
        for p in range(4):
            q = call * 4 + p
            for k in range(6):
                J[k, p] = np.float32(0.25) * np.float32((q * 7 + k * 3) % 11) - np.float32(1.0)
            res[p] = float(q % 5) - 2.0
            wgt[p] = 0.5 + 0.125 * (q % 3)
        O.ls_update4(o, J, res, wgt, quirk_plus6=True)
    A, bvec, err, cnt = O.ls_finish(o, divide=True)
    sse = lines["LSSSE"]
    got = np.array([float(v) for v in sse[:5]], np.float32)
    want = np.array([A[0, 0], A[1, 4], A[5, 5], bvec[3], err], np.float32)
    assert int(sse[5]) == cnt == 12
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6), (got, want)      # the bar of test_ls_accumulate_matches_oracle_ls
    # FastEstimatePose = EstimatePose's terms under the 4 -> 0 / 50 / gain 50 schedule
    fast = lines["FAST"]
    fp = dict(n_levels=5, first_level=4, last_level=0, max_iters=50, early_exit=1, gain=50.0)
    sf, pose_fast, tr_fast = O.align_pair(O.default_params(w, h, *intr, **fp), ref, tgt, want_trace=True)
    assert sf == 0 and int(fast[7]) == len(tr_fast)
    assert np.array_equal(np.array([float(v) for v in fast[:7]], np.float32), pose_fast)
    # a frame whose slot went to another frame: told so (slot -1), EstimatePose refuses until ApplyGradient ran again,
    # then reproduces the first pose bit for bit
    assert lines["EVICT"] == ["-1", "1", "1"]


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
    # the multi-GPU mode's path with the one device of the box: RCCL communicator, all-gather on the context's stream
    out = subprocess.run([exe, "--pairs", "12", "--unique", "3", "--width", "160", "--height", "96", "--levels", "3", "--steps", "2",
                          "--warmup", "1", "--gpus", "1", "--rccl"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["gathered_blocks_match"] is True and d["tiled_pairs_identical"] and d["n_gpus"] == 1
    out = subprocess.run([exe, "--gpus", "64", "--pairs", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "device(s) visible" in out.stderr
