"""Generates tests/golden/*.npz from the CPU oracle (oracle/uwt_oracle.c) on seeded synthetic inputs.

The reference repo holds no golden vectors, fixtures or known-answer tests for this path
and cannot be built here, so these fixtures pin the ORACLE's behaviour (regression) and give the GPU parity
tests committed expected values.  Independent cross-checks of the oracle's math (scipy expm, numpy inv,
scipy.ndimage correlate) live in tests/test_oracle.py.

Every arithmetic-dependent vector exists twice: under its plain key for the default set (OpenCV's generic paths,
O.ARITH_OPENCV) and under "legacy_<key>" for the legacy set (tests/conftest.py::GoldenView picks by the active set).

Run from the repo root:  python tests/golden/make_golden.py [output directory, default: tests/golden]
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

synth = importlib.import_module("uw-slam_amd.synth")
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.abspath(__file__))   # (another directory: tools/sanitize_cpu.sh)


def pack_trace(tr):
    keys = ["level", "iter", "n_valid", "exited", "sum_r2", "error"]
    d = {k: np.array([t[k] for t in tr]) for k in keys}
    for k in ["A", "b", "delta", "pose"]:
        d[k] = np.stack([t[k] for t in tr]) if tr else np.zeros((0,))
    return d


def pair_case(name, w, h, intr, seed, over, with_depth=False, z=1.0, max_t=0.01, max_deg=0.5):
    fx, fy, cx, cy = intr
    ref, tgt, depth, R, t = synth.render_pair(w, h, fx, fy, cx, cy, seed, z=z, max_t=max_t, max_deg=max_deg,
                                              with_depth=with_depth)
    d = dict(ref=ref, tgt=tgt, intr=np.array(intr, np.float32),
             over_keys=np.array(list(over.keys())), over_vals=np.array([float(v) for v in over.values()]))
    if with_depth:
        d["depth"] = depth
    for arith, prefix in ((O.ARITH_OPENCV, ""), (O.ARITH_LEGACY, "legacy_")):
        p = O.default_params(w, h, fx, fy, cx, cy, arith=arith, **over)
        if with_depth:
            p.has_depth = 1
        st, pose, tr = O.align_pair(p, ref, tgt, depth if with_depth else None, want_trace=True)
        d.update({prefix + "trace_" + k: v for k, v in pack_trace(tr).items()})
        d[prefix + "pose"] = pose
        d[prefix + "status"] = np.int32(st)
        print(name, O.ARITH_NAMES[arith], "status", st, "rows", len(tr), "pose", pose)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)


def main():
    rng = np.random.default_rng(1234)
    # (1) SE3 exp / mul / handoff vectors
    xis = np.concatenate([rng.normal(0, 0.05, (24, 6)), rng.normal(0, 1e-7, (4, 6)), np.zeros((1, 6)),
                          rng.normal(0, 1.0, (3, 6))]).astype(np.float32)
    exps = np.stack([O.se3_exp(x) for x in xis])
    muls = np.stack([O.se3_mul(exps[i], exps[(i + 1) % len(exps)]) for i in range(len(exps))])
    hand = np.stack([O.se3_handoff(e, 0) for e in exps])
    hand_t = np.stack([O.se3_handoff(e, 1) for e in exps])
    mats = np.stack([O.se3_matrix(e) for e in exps])
    np.savez_compressed(os.path.join(OUT, "se3.npz"), xi=xis, exp=exps, mul=muls, handoff=hand, handoff_t=hand_t, mat=mats)

    # (2) 6x6 inverse incl. a singular case
    As = []
    for i in range(8):
        M = rng.normal(0, 1, (40, 6)).astype(np.float32)
        M[:, 3:] *= 100.0  # badly scaled like the tracker's A
        As.append((M.T @ M).astype(np.float32))
    S = As[0].copy(); S[:, 2] = 0; S[2, :] = 0
    As.append(S)
    As = np.stack(As)
    inv = np.stack([O.inv6(a)[0] for a in As])
    ok = np.array([O.inv6(a)[1] for a in As])
    bs = rng.normal(0, 1, (len(As), 6)).astype(np.float32)
    O.set_arith(O.ARITH_OPENCV)   # "A.inv() * b" = cv::solve: LU on the right-hand side
    delta = np.stack([O.solve_delta(a, b) for a, b in zip(As, bs)])
    O.set_arith(O.ARITH_LEGACY)   # the inverse formed, then multiplied
    delta_legacy = np.stack([O.solve_delta(a, b) for a, b in zip(As, bs)])
    O.set_arith(O.ARITH_OPENCV)
    np.savez_compressed(os.path.join(OUT, "inv6.npz"), A=As, inv=inv, ok=ok, b=bs, delta=delta, legacy_delta=delta_legacy)

    # (3) frame pairs + per-iteration traces
    small = (64.0, 64.0, 31.5, 23.5)
    pair_case("pair_64x48_ref", 64, 48, small, 11,
              dict(n_levels=3, first_level=2, last_level=0, max_iters=50, early_exit=1), max_t=0.02, max_deg=1.0)
    pair_case("pair_64x48_fixed", 64, 48, small, 12,
              dict(n_levels=3, first_level=2, last_level=0, max_iters=6, early_exit=0), max_t=0.02, max_deg=1.0)
    mid = (131.25, 131.25, 79.5, 47.5)
    pair_case("pair_160x96_ref5", 160, 96, mid, 13, dict())  # reference defaults: 5 levels, 4->1, 50 it, early exit
    pair_case("pair_160x96_ref5_depth", 160, 96, mid, 17, dict(), with_depth=True, z=1.05)  # reference constants + depth plane
    pair_case("pair_160x96_fixed", 160, 96, mid, 14,
              dict(n_levels=4, first_level=3, last_level=0, max_iters=10, early_exit=0))
    pair_case("pair_160x96_depth", 160, 96, mid, 15,
              dict(n_levels=4, first_level=3, last_level=0, max_iters=5, early_exit=0), with_depth=True, z=1.1)
    pair_case("pair_160x96_features", 160, 96, mid, 16,  # EstimatePoseFeatures constants (Tracker.cpp:634-640, 834, 856)
              dict(n_levels=5, first_level=0, last_level=0, max_iters=10, early_exit=1, gain=1.0, z_factor=0.002,
                   handoff_scale_t=1))

    # (4) stage KATs on a tiny image
    img = rng.integers(0, 256, (12, 16), dtype=np.uint8)
    gx, gy = O.scharr3(img)
    dep = rng.integers(0, 65536, (12, 16)).astype(np.uint16)
    np.savez_compressed(os.path.join(OUT, "stages.npz"), img=img, half=O.halve_u8(img), gx=gx, gy=gy,
                        mag=O.gradient_mag(gx, gy), dep=dep, dep_half=O.halve_u16(dep))

    # (5) LS traces
    J = rng.normal(0, 10, (16, 6)).astype(np.float32)
    r = rng.integers(-50, 50, 16).astype(np.float32)
    w = rng.uniform(0.1, 1, 16).astype(np.float32)
    ls = O.ls_new()
    for i in range(16):
        O.ls_update(ls, J[i], r[i], w[i])
    A1, b1, e1, n1 = O.ls_finish(ls, True)
    ls = O.ls_new()
    for i in range(0, 16, 4):
        O.ls_update4(ls, J[i:i + 4].T.copy(), r[i:i + 4], w[i:i + 4], True)
    A4, b4, e4, n4 = O.ls_finish(ls, False)
    np.savez_compressed(os.path.join(OUT, "ls.npz"), J=J, r=r, w=w, A_scalar=A1, b_scalar=b1, err_scalar=e1,
                        n_scalar=n1, A_sse=A4, b_sse=b4, err_sse=e4, n_sse=n4)


if __name__ == "__main__":
    main()
