"""Sequential tracking of a recorded sequence through the batched GPU path (BASELINE configs 1, 2, 5).

Every alignment starts from identity (src/Tracker.cpp:385), so tracking frame i against frame i+1 for a whole
sequence is one batch: pair i = (slot i, slot i+1).  This module is host plumbing: directory listing in the
reference's order (System::AddLists, src/System.cpp:290-350), image / depth reading, centre crop to a size the
pyramid accepts, chunked upload, and the trajectory hand-off to uwt_accumulate_trajectory.
"""
import os

import numpy as np

from . import capi


def list_sorted(directory):
    """src/System.cpp:296-310: every directory entry, sorted alphabetically ('.' and '..' dropped)."""
    names = sorted(n for n in os.listdir(directory) if n not in (".", ".."))
    return [os.path.join(directory, n) for n in names]


def load_gray(path):
    """imread(path, CV_LOAD_IMAGE_GRAYSCALE) (src/System.cpp:228).  8-bit grey files are returned as stored; colour
    files go through PIL's ITU-R 601 luma, which can differ from OpenCV's fixed-point conversion by one grey level."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("L"), dtype=np.uint8)


def load_depth(path):
    """imread(path, -1) of a 16-bit depth PNG (src/System.cpp:243)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im, dtype=np.uint16)


def centre_crop(img, w, h):
    H, W = img.shape
    if W < w or H < h:
        raise ValueError("image %dx%d smaller than crop %dx%d" % (W, H, w, h))
    x0, y0 = (W - w) // 2, (H - h) // 2
    return np.ascontiguousarray(img[y0:y0 + h, x0:x0 + w]), x0, y0


class SequenceTracker:
    """Tracks consecutive frame pairs of a sequence in chunks of `chunk` pairs, streamed: two slot ranges on the device
    and two page-locked staging blocks on the host alternate, so that chunk k + 1 is staged and crosses PCIe
    (uwt_upload_frames_async, copy stream) while chunk k is aligned (uwt_track_batch_host_async, context stream) — the
    per-frame flow of System::AddFrame + System::Tracking (src/System.cpp:225-251, 193-223) with the upload overlapped.
    The frame shared by two chunks is uploaded with both."""

    def __init__(self, width, height, fx, fy, cx, cy, depth=False, chunk=256, device=0, **params):
        self.w, self.h, self.depth, self.chunk = width, height, depth, chunk
        over = dict(max_frames=2 * (chunk + 1), max_pairs=chunk, has_depth=int(depth), device=device)
        over.update(params)
        self.ctx = capi.Context(capi.default_params(width, height, fx, fy, cx, cy, **over))
        self._gray = [capi.pinned_empty((chunk + 1, height, width), np.uint8) for _ in range(2)]
        self._depth = [capi.pinned_empty((chunk + 1, height, width), np.uint16) for _ in range(2)] if depth else [None, None]
        self._poses = [capi.pinned_empty((chunk, 7), np.float32) for _ in range(2)]
        self._stats = [capi.pinned_empty((chunk, 4), np.int32) for _ in range(2)]

    def _stage_and_upload(self, k, frames, depths, start, m):
        b = k % 2
        for i in range(m + 1):                                  # into page-locked memory, straight from the reader's arrays
            self._gray[b][i] = frames[start + i]
            if self.depth:
                self._depth[b][i] = depths[start + i]
        self.ctx.upload_frames_async(b * (self.chunk + 1), self._gray[b][:m + 1],
                                     self._depth[b][:m + 1] if self.depth else None)

    def track(self, frames, depths=None):
        """frames: sequence of h x w uint8 arrays (depths: matching uint16).  Returns (poses [n-1, 7], stats list):
        pose i is previous_frame->rigid_transformation_ for the pair (frame i, frame i+1)."""
        frames = list(frames)
        depths = list(depths) if depths is not None else None
        n = len(frames)
        chunks = []                                             # (first frame, pairs)
        start = 0
        while start < n - 1:
            m = min(self.chunk, n - 1 - start)                  # pairs of this chunk use frames start .. start + m
            chunks.append((start, m))
            start += m
        poses = np.zeros((max(n - 1, 0), 7), np.float32)
        stats = []
        tickets = {}

        def collect(k):
            self.ctx.wait_ticket(tickets.pop(k))
            s0, m = chunks[k]
            poses[s0:s0 + m] = self._poses[k % 2][:m]
            for r in self._stats[k % 2][:m]:
                stats.append(dict(status=int(r[0]), iterations=int(r[1]), n_valid=int(r[2]),
                                  error=float(np.asarray(r[3:4]).view(np.float32)[0])))

        for k in range(min(2, len(chunks))):
            self._stage_and_upload(k, frames, depths, *chunks[k])
        for k, (s0, m) in enumerate(chunks):
            base = (k % 2) * (self.chunk + 1)
            ref = base + np.arange(m, dtype=np.int32)
            tickets[k] = self.ctx.track_batch_host_async(base, m + 1, ref, ref + 1, self._poses[k % 2], self._stats[k % 2])
            if k >= 1:
                collect(k - 1)                                  # frees the other staging block and slot range ...
                if k + 1 < len(chunks):
                    self._stage_and_upload(k + 1, frames, depths, *chunks[k + 1])   # ... for the chunk after this one
        if chunks:
            collect(len(chunks) - 1)
        return poses, stats

    def trajectory(self, poses, start_pose=None, reference_visualiser=False, scan=False):
        """Visualizer::UpdateMessages accumulation (src/Visualizer.cpp:304-325); reference_visualiser reproduces the
        x40 translation scale and the (-z, -x, -y) axis permutation (sequential, bit-exact).  scan=True takes the plain
        SE(3) prefix product as a parallel scan (long trajectories; equal to the sequential form to float rounding)."""
        if reference_visualiser:
            return self.ctx.accumulate_trajectory(poses, start_pose, 40.0, True)
        return self.ctx.accumulate_trajectory(poses, start_pose, 1.0, False, scan=scan)

    def close(self):
        self.ctx.close()


def ate_rmse(est_xyz, gt_xyz):
    """Absolute trajectory error after a rigid (Horn / Umeyama without scale) alignment of est onto gt."""
    est = np.asarray(est_xyz, np.float64)
    gt = np.asarray(gt_xyz, np.float64)
    mu_e, mu_g = est.mean(0), gt.mean(0)
    H = (est - mu_e).T @ (gt - mu_g)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    aligned = (R @ (est - mu_e).T).T + mu_g
    return float(np.sqrt(((aligned - gt) ** 2).sum(1).mean()))


def rpe_translation(est_rel_t, gt_rel_t):
    """Relative pose error (translation part) between per-pair estimated and ground-truth translations."""
    d = np.asarray(est_rel_t, np.float64) - np.asarray(gt_rel_t, np.float64)
    return float(np.sqrt((d ** 2).sum(1).mean()))
