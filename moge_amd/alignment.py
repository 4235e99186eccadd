"""Optimal-alignment solvers of the evaluation path on the MI355X: the host-side mirror of the reference's `moge/utils/alignment.py`
(same function names, arguments, return values), calling the HIP kernels of `csrc/alignment.hip` through the C ABI (`moge_align_*`).

    from moge_amd.alignment import align_points_scale_xyz_shift      # instead of moge.utils.alignment
    scale, shift = align_points_scale_xyz_shift(pred_points_lr, gt_points_lr, 1 / gt_points_lr.norm(dim=-1))     # test/metrics.py:264

Every tensor must live on the GPU (`cuda`); there is no CPU path here (the library raises without a device).  The reference builds an
(anchors, n, 3) tensor per call for the affine solvers; the kernels subtract the anchor while loading, so the only temporaries are the
per-anchor results.

Not mirrored: the truncated objective (`trunc is not None`, alignment.py:91-144) and `align_depth_affine_irls` (alignment.py:214-226) - both
serve the training losses (train/losses.py), out of scope here; passing `trunc` raises NotImplementedError.
Differentiability: the reference returns `scale` / `shift` recomputed from the selected samples so that gradients flow to them
(alignment.py:199-209); this mirror serves evaluation (test/metrics.py runs under no_grad) and returns plain tensors."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L

MAX_ROW = 15360          # residuals per row (csrc/alignment.hip: a row is sorted inside one CU's LDS)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("moge_amd.alignment works on GPU tensors only (no CPU path)")


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _no_trunc(trunc):
    if trunc is not None:
        raise NotImplementedError("the truncated objective (alignment.py:91-144) is used by the training losses only and is not built")


def align(x: torch.Tensor, y: torch.Tensor, w: torch.Tensor, trunc=None, eps: float = 1e-7) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """alignment.py:52-89: min_a sum_i w_i |a x_i - y_i| per row of the broadcast (..., n) inputs -> a (...), loss (...), index (...) (int64)."""
    _no_trunc(trunc)
    _need_cuda(x, y, w)
    x, y, w = torch.broadcast_tensors(x, y, w)
    bshape, n = x.shape[:-1], x.shape[-1]
    x, y, w = (t.reshape(-1, n).float().contiguous() for t in (x, y, w))
    rows = x.shape[0]
    a = torch.empty(rows, device=x.device, dtype=torch.float32)
    loss = torch.empty_like(a)
    index = torch.empty(rows, device=x.device, dtype=torch.int32)
    L.check(L.lib.moge_align_l1(_p(x), _p(y), _p(w), rows, n, eps, _p(a), _p(loss), _p(index), _stream()))
    return a.reshape(bshape), loss.reshape(bshape), index.long().reshape(bshape)


def _anchor_search(src: torch.Tensor, tgt: torch.Tensor, weight: torch.Tensor, comp_mask: int):
    """src / tgt (B, n, d), weight (B, n): one solve per sample with weight > 0 (alignment.py:184 / :269 / :324), then the best anchor per batch
    element (alignment.py:197 / :284 / :339).  -> anchor sample (B,), solution element (B,) in [0, n*d)"""
    B, n, d = src.shape
    ab, an = torch.where(weight > 0)
    rows = ab.numel()
    if rows == 0:
        raise ValueError("no sample with weight > 0")
    rb, rk = ab.int().contiguous(), an.int().contiguous()
    scale = torch.empty(rows, device=src.device, dtype=torch.float32)
    loss = torch.empty_like(scale)
    index = torch.empty(rows, device=src.device, dtype=torch.int32)
    L.check(L.lib.moge_align_l1_anchored(_p(src), _p(tgt), _p(weight), n, d, comp_mask, _p(rb), _p(rk), rows, 1e-7, _p(scale), _p(loss), _p(index), _stream()))
    min_loss = torch.empty(B, device=src.device, dtype=torch.float32)
    min_row = torch.empty(B, device=src.device, dtype=torch.int32)
    L.check(L.lib.moge_align_select(_p(loss), _p(rb), rows, B, _p(min_loss), _p(min_row), _stream()))
    sel = min_row.long()
    if bool((sel < 0).any()):
        raise ValueError("a batch element has no sample with weight > 0")       # the reference indexes with -1 here (last anchor of the batch)
    return an[sel], index.long()[sel]


def align_depth_scale(depth_src: torch.Tensor, depth_tgt: torch.Tensor, weight: Optional[torch.Tensor], trunc=None):
    """alignment.py:149-160"""
    return align(depth_src, depth_tgt, weight, trunc)[0]


def align_depth_affine(depth_src: torch.Tensor, depth_tgt: torch.Tensor, weight: Optional[torch.Tensor], trunc=None):
    """alignment.py:163-212: (..., n) -> scale (...), shift (...)"""
    _no_trunc(trunc)
    _need_cuda(depth_src, depth_tgt, weight)
    bshape, n = depth_src.shape[:-1], depth_src.shape[-1]
    src, tgt, w = (t.reshape(-1, n).float().contiguous() for t in (depth_src, depth_tgt, weight))
    i1, i2 = _anchor_search(src[..., None], tgt[..., None], w, 0b1)
    t1, s1 = tgt.gather(1, i1[:, None])[:, 0], src.gather(1, i1[:, None])[:, 0]
    t2, s2 = tgt.gather(1, i2[:, None])[:, 0], src.gather(1, i2[:, None])[:, 0]
    scale = (t2 - t1) / torch.where(s2 != s1, s2 - s1, torch.full_like(s1, 1e-7))          # :206
    shift = t1 - scale * s1                                                                 # :207
    return scale.reshape(bshape), shift.reshape(bshape)


def align_points_scale(points_src: torch.Tensor, points_tgt: torch.Tensor, weight: Optional[torch.Tensor], trunc=None):
    """alignment.py:228-243: (..., n, 3) -> scale (...)"""
    return align(points_src.flatten(-2), points_tgt.flatten(-2), weight[..., None].expand_as(points_src).flatten(-2), trunc)[0]


def _points_anchor_solve(points_src, points_tgt, weight, comp_mask: int):
    _need_cuda(points_src, points_tgt, weight)
    bshape, n = points_src.shape[:-2], points_src.shape[-2]
    src, tgt, w = points_src.reshape(-1, n, 3).float().contiguous(), points_tgt.reshape(-1, n, 3).float().contiguous(), weight.reshape(-1, n).float().contiguous()
    B = src.shape[0]
    k, i2 = _anchor_search(src, tgt, w, comp_mask)
    i1 = k * 3 + i2 % 3                                                                     # :288 / :342
    m = torch.tensor([(comp_mask >> c) & 1 for c in range(3)], device=src.device, dtype=src.dtype)
    src_a, tgt_a = src * m, tgt * m                                                         # :290-291 (zeros where the anchor is not subtracted)
    t1, s1 = tgt_a.reshape(B, -1).gather(1, i1[:, None])[:, 0], src_a.reshape(B, -1).gather(1, i1[:, None])[:, 0]
    t2, s2 = tgt.reshape(B, -1).gather(1, i2[:, None])[:, 0], src.reshape(B, -1).gather(1, i2[:, None])[:, 0]
    scale = (t2 - t1) / torch.where(s2 != s1, s2 - s1, torch.ones_like(s1))                 # :295 / :348
    rows = torch.arange(B, device=src.device)
    shift = tgt_a[rows, k] - scale[:, None] * src_a[rows, k]                                # :296 / :349
    return scale.reshape(bshape), shift.reshape(*bshape, 3)


def align_points_scale_z_shift(points_src: torch.Tensor, points_tgt: torch.Tensor, weight: Optional[torch.Tensor], trunc=None):
    """alignment.py:246-299: shared xyz scale + shift along z."""
    _no_trunc(trunc)
    return _points_anchor_solve(points_src, points_tgt, weight, 0b100)


def align_points_scale_xyz_shift(points_src: torch.Tensor, points_tgt: torch.Tensor, weight: Optional[torch.Tensor], trunc=None, max_iters: int = 30, eps: float = 1e-6):
    """alignment.py:302-354: shared xyz scale + xyz shift (max_iters / eps are unused in the reference as well)."""
    _no_trunc(trunc)
    return _points_anchor_solve(points_src, points_tgt, weight, 0b111)


def align_points_z_shift(points_src: torch.Tensor, points_tgt: torch.Tensor, weight: Optional[torch.Tensor], trunc=None, max_iters: int = 30, eps: float = 1e-6):
    """alignment.py:357-376"""
    shift = align(torch.ones_like(points_src[..., 2]), points_tgt[..., 2] - points_src[..., 2], weight, trunc)[0]
    return torch.stack([torch.zeros_like(shift), torch.zeros_like(shift), shift], dim=-1)


def align_points_xyz_shift(points_src: torch.Tensor, points_tgt: torch.Tensor, weight: Optional[torch.Tensor], trunc=None, max_iters: int = 30, eps: float = 1e-6):
    """alignment.py:379-396"""
    return align(torch.ones_like(points_src).swapaxes(-2, -1), (points_tgt - points_src).swapaxes(-2, -1), weight[..., None, :], trunc)[0]


def align_affine_lstsq(x: torch.Tensor, y: torch.Tensor, w: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """alignment.py:399-415: least-squares (a, b) of sqrt(w) x a + b ~ sqrt(w) y per row of (..., N)."""
    _need_cuda(x, y, w)
    bshape, n = x.shape[:-1], x.shape[-1]
    xs, ys = x.reshape(-1, n).float().contiguous(), y.reshape(-1, n).float().contiguous()
    ws = w.reshape(-1, n).float().contiguous() if w is not None else None
    rows = xs.shape[0]
    a = torch.empty(rows, device=x.device, dtype=torch.float32)
    b = torch.empty_like(a)
    L.check(L.lib.moge_align_lstsq(_p(xs), _p(ys), _p(ws), rows, n, _p(a), _p(b), _stream()))
    return a.reshape(bshape), b.reshape(bshape)
