#!/usr/bin/env python3
"""Time the alignment solvers at the evaluation size (64 x 64 samples) on the GPU: kernel milliseconds per call (HIP events on the launch
stream), next to the torch implementation of the same algorithm when a reference checkout is importable (never on the GPU box)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from moge_amd import alignment as A


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    out = {}
    for B, n in ((1, 4096), (8, 4096), (1, 1024)):
        gt = torch.rand(B, n, 3, device="cuda", generator=g) * 4 + 0.5
        pred = (gt - 0.1) / 1.7 + 0.01 * torch.randn(B, n, 3, device="cuda", generator=g)
        w = 1.0 / gt.norm(dim=-1)
        out[f"B{B}_n{n}"] = {
            "points_scale_xyz_shift_ms": round(timed(lambda: A.align_points_scale_xyz_shift(pred, gt, w)), 3),
            "depth_affine_ms": round(timed(lambda: A.align_depth_affine(pred[..., 2], gt[..., 2], w)), 3),
            "points_scale_ms": round(timed(lambda: A.align_points_scale(pred, gt, w)), 3),
            "rows_x_residuals": [B * n, 3 * n],
        }
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
