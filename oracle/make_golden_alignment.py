"""Golden vectors for the alignment solvers: inputs + outputs of the REFERENCE functions (moge/utils/alignment.py, run on CPU here).

    python -m oracle.make_golden_alignment            # writes tests/golden/align_*.npz

Needs /root/reference (this container only).  utils3d / cv2 are stubbed exactly as in oracle/make_golden.py - alignment.py imports utils3d
but never calls it.  Committed together with the fixtures it writes; tests/ only read the .npz files."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from .make_golden import GOLDEN_DIR, install_stubs


def scene(rng, B, n, outliers=0.1, zero_w=0.2):
    """A plausible evaluation sample: ground-truth points in front of the camera, a prediction that is an affine transform of them plus noise
    and a few gross outliers, weights 1 / |gt| with a share of zeros (masked-out samples)."""
    gt = np.stack([rng.uniform(-2, 2, (B, n)), rng.uniform(-1.5, 1.5, (B, n)), rng.uniform(0.5, 8, (B, n))], -1).astype(np.float32)
    scale = rng.uniform(0.3, 3, (B, 1, 1)).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, (B, 1, 3)).astype(np.float32)
    pred = (gt - shift) / scale + rng.normal(0, 0.01, gt.shape).astype(np.float32)
    bad = rng.random((B, n)) < outliers
    pred[bad] += rng.normal(0, 1.0, (int(bad.sum()), 3)).astype(np.float32)
    w = (1.0 / np.linalg.norm(gt, axis=-1)).astype(np.float32)
    w[rng.random((B, n)) < zero_w] = 0
    return pred, gt, w


def main():
    install_stubs()
    from moge.utils import alignment as A          # the reference
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    rng = np.random.default_rng(20260922)
    t = torch.from_numpy

    # ---- the 1-D solve itself -------------------------------------------------------------------------------------------------------
    x = rng.normal(0, 1, (7, 33)).astype(np.float32)
    x[:, ::11] = 0                                   # zeros: sign(x) = 0 rows of the problem
    y = (1.7 * x + rng.normal(0, 0.3, x.shape)).astype(np.float32)
    w = rng.uniform(0, 2, x.shape).astype(np.float32)
    w[:, 5] = 0
    a, loss, idx = A.align(t(x), t(y), t(w))
    np.savez(os.path.join(GOLDEN_DIR, "align_l1_small.npz"), x=x, y=y, w=w, a=a.numpy(), loss=loss.numpy(), index=idx.numpy())

    # exactly representable: small integers, every partial sum exact in fp32 -> the index itself is pinned
    x = rng.integers(1, 6, (5, 40)).astype(np.float32)
    y = (x * rng.integers(-8, 9, (5, 40))).astype(np.float32)
    w = rng.integers(0, 4, (5, 40)).astype(np.float32)
    a, loss, idx = A.align(t(x), t(y), t(w))
    np.savez(os.path.join(GOLDEN_DIR, "align_l1_exact.npz"), x=x, y=y, w=w, a=a.numpy(), loss=loss.numpy(), index=idx.numpy())

    # a long row (the size one anchor row has in test/metrics.py: 3 * 64 * 64 residuals)
    pred, gt, wt = scene(rng, 1, 4096)
    xs, ys, ws = pred.reshape(1, -1), gt.reshape(1, -1), np.repeat(wt, 3, axis=-1)
    a, loss, idx = A.align(t(xs), t(ys), t(ws))
    np.savez(os.path.join(GOLDEN_DIR, "align_l1_long.npz"), x=xs, y=ys, w=ws, a=a.numpy(), loss=loss.numpy(), index=idx.numpy())

    # ---- the solvers test/metrics.py calls (and the z-shift variants beside them) ------------------------------------------------------
    for name, (B, n) in {"align_solvers_small": (3, 150), "align_solvers_lr": (1, 1024), "align_solvers_full": (1, 4096)}.items():
        pred, gt, wt = scene(rng, B, n)
        out = dict(pred=pred, gt=gt, w=wt)
        P, G, W = t(pred), t(gt), t(wt)
        out["depth_scale"] = A.align_depth_scale(P[..., 2], G[..., 2], W).numpy()
        s, sh = A.align_depth_affine(P[..., 2], G[..., 2], W)
        out["depth_affine_scale"], out["depth_affine_shift"] = s.numpy(), sh.numpy()
        out["points_scale"] = A.align_points_scale(P, G, W).numpy()
        s, sh = A.align_points_scale_z_shift(P, G, W)
        out["points_scale_z_shift_scale"], out["points_scale_z_shift_shift"] = s.numpy(), sh.numpy()
        s, sh = A.align_points_scale_xyz_shift(P, G, W)
        out["points_scale_xyz_shift_scale"], out["points_scale_xyz_shift_shift"] = s.numpy(), sh.numpy()
        out["points_z_shift"] = A.align_points_z_shift(P, G, W).numpy()
        out["points_xyz_shift"] = A.align_points_xyz_shift(P, G, W).numpy()
        a_, b_ = A.align_affine_lstsq(P[..., 2], 1.0 / G[..., 2])
        out["lstsq_a"], out["lstsq_b"] = a_.numpy(), b_.numpy()
        a_, b_ = A.align_affine_lstsq(P[..., 2], 1.0 / G[..., 2], W + 0.1)
        out["lstsq_w_a"], out["lstsq_w_b"] = a_.numpy(), b_.numpy()
        np.savez(os.path.join(GOLDEN_DIR, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items() if k not in ("pred", "gt", "w")})


if __name__ == "__main__":
    sys.exit(main())
