"""uw-slam_amd — MI355X-native (gfx950) direct SE(3) tracking path of UW-SLAM behind a C ABI.

Layout:  csrc/ (HIP kernels + C ABI → libuwt_hip.so), capi.py (ctypes binding), tracker.py (host-side mirror of the
reference's Tracker / LS class surface), synth.py (synthetic inputs), dist.py (multi-GPU sharding of pairs).
The directory name carries a hyphen; import it with importlib.import_module("uw-slam_amd").
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_native(force=False):
    """hipcc --offload-arch=gfx950 build of libuwt_hip.so (recipe: csrc/Makefile)."""
    cmd = ["make", "-j%d" % max(1, min(8, os.cpu_count() or 1)), "-C", os.path.join(_HERE, "csrc")]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return os.path.join(_HERE, "libuwt_hip.so")
