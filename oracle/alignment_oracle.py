"""CPU restatement (numpy, float32) of the reference's optimal-alignment solvers.  TEST INFRASTRUCTURE ONLY: imported by tests/ and by
oracle/make_golden_alignment.py, never by the product (moge_amd/alignment.py calls the HIP kernels through the C ABI).

What is restated (`/root/reference/moge/utils/alignment.py`):
  align(x, y, w)                      alignment.py:52-89    min_a sum_i w_i |a x_i - y_i|   (trunc=None branch: the one test/metrics.py uses)
  scatter_min                         alignment.py:13-20    per-batch minimum over the anchor rows + which row
  align_depth_scale                   alignment.py:149-160
  align_depth_affine                  alignment.py:163-212
  align_points_scale                  alignment.py:228-243
  align_points_scale_z_shift          alignment.py:246-299
  align_points_scale_xyz_shift        alignment.py:302-354
  align_points_z_shift                alignment.py:357-376
  align_points_xyz_shift              alignment.py:379-396
  align_affine_lstsq                  alignment.py:399-415
Not restated: the truncated objective (trunc is not None, alignment.py:91-144) and align_depth_affine_irls - both are used by the training
losses only (train/losses.py), which SURVEY.md 8 marks out of scope.

Parity status: PINNED - tests/golden/align_*.npz hold inputs and outputs of the reference functions themselves, run on CPU by
oracle/make_golden_alignment.py; tests/test_alignment_oracle.py checks this file against them.

Numerics: the solution of the 1-D problem is one of the ratios y_i / x_i; which one is decided by the sign change of a prefix sum taken in
sorted order.  torch (pairwise / parallel sums) and numpy (sequential cumsum) round those sums differently, so on near-ties the chosen
index may differ between implementations while the objective value agrees to rounding - the tests therefore compare OBJECTIVE VALUES
tightly and solutions with a tolerance, and demand identical indices only on the exactly-representable fixtures."""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _f(a):
    return np.ascontiguousarray(np.asarray(a, dtype=F32))


def align(x, y, w, eps: float = 1e-7):
    """alignment.py:52-89 (trunc=None).  x, y, w broadcastable to (..., n).  -> a (...), loss (...), index (...)"""
    x, y, w = np.broadcast_arrays(_f(x), _f(y), _f(w))
    sign = np.sign(x).astype(F32)                                            # :71
    x, y = x * sign, y * sign                                                 # :72
    ratio = (y / np.maximum(x, F32(eps))).astype(F32)                         # :73
    order = np.argsort(ratio, axis=-1, kind="stable")                         # :74
    ratio_s = np.take_along_axis(ratio, order, axis=-1)
    wx = np.take_along_axis((x * w).astype(F32), order, axis=-1)              # :76
    total = wx.sum(axis=-1, keepdims=True, dtype=np.float64)
    deriv = 2.0 * np.cumsum(wx, axis=-1, dtype=np.float64) - total            # :77 (float64 here: the sign change is decided exactly)
    ge = deriv >= 0
    search = np.where(ge.any(axis=-1), ge.argmax(axis=-1), ratio.shape[-1] - 1)   # :78 searchsorted(..., 0, 'left').clamp_max(n - 1)
    a = np.take_along_axis(ratio_s, search[..., None], axis=-1)[..., 0]       # :80
    index = np.take_along_axis(order, search[..., None], axis=-1)[..., 0]     # :81
    loss = (w * np.abs(a[..., None] * x - y)).sum(axis=-1, dtype=np.float64).astype(F32)      # :82
    return a.astype(F32), loss, index.astype(np.int64)


def objective(a, x, y, w):
    """sum_i w_i |a x_i - y_i| in float64 (what the tests compare)."""
    x, y, w = np.broadcast_arrays(np.asarray(x, np.float64), np.asarray(y, np.float64), np.asarray(w, np.float64))
    return (w * np.abs(np.asarray(a, np.float64)[..., None] * x - y)).sum(axis=-1)


def scatter_min(size: int, index, src):
    """alignment.py:13-20 along dim 0: minimum of src per target slot, and the position (LAST one on ties: the reference's indexed assignment
    runs in order on CPU) that attains it; slots without entries: (+inf, -1)."""
    src = _f(src)
    index = np.asarray(index, np.int64)
    minimum = np.full(size, np.inf, F32)
    np.minimum.at(minimum, index, src)
    where = np.nonzero(src == minimum[index])[0]
    indices = np.full(size, -1, np.int64)
    indices[index[where]] = where                                             # later entries overwrite earlier ones
    return minimum, indices


def align_depth_scale(depth_src, depth_tgt, weight):
    return align(depth_src, depth_tgt, weight)[0]                             # alignment.py:158


def _anchored(src, tgt, weight, comp_mask, chunk=256):
    """Shared body of the anchor searches: src / tgt (B, n, d), weight (B, n); every element with weight > 0 is an anchor, subtracted from
    the components selected by comp_mask; one 1-D solve per anchor over the flattened (n * d) residuals.
    -> anchors_b, anchors_n, scale, loss, index (per anchor)"""
    B, n, d = src.shape
    ab, an = np.nonzero(weight > 0)                                           # :184 / :269 / :324
    m = np.asarray(comp_mask, F32)
    scale = np.empty(ab.size, F32); loss = np.empty(ab.size, F32); index = np.empty(ab.size, np.int64)
    for s in range(0, ab.size, chunk):
        b, k = ab[s:s + chunk], an[s:s + chunk]
        xs = src[b] - (src[b, k] * m)[:, None, :]                             # :191 / :274 / :331
        ys = tgt[b] - (tgt[b, k] * m)[:, None, :]
        ws = np.broadcast_to(weight[b][:, :, None], xs.shape)
        sc, lo, ix = align(xs.reshape(len(b), -1), ys.reshape(len(b), -1), ws.reshape(len(b), -1))
        scale[s:s + chunk], loss[s:s + chunk], index[s:s + chunk] = sc, lo, ix
    return ab, an, scale, loss, index


def align_depth_affine(depth_src, depth_tgt, weight):
    """alignment.py:163-212.  (..., n) -> scale (...), shift (...)"""
    depth_src, depth_tgt, weight = _f(depth_src), _f(depth_tgt), _f(weight)
    bshape, n = depth_src.shape[:-1], depth_src.shape[-1]
    src, tgt, w = depth_src.reshape(-1, n), depth_tgt.reshape(-1, n), weight.reshape(-1, n)
    B = src.shape[0]
    ab, an, _, loss, index = _anchored(src[..., None], tgt[..., None], w, [1.0])
    _, ia = scatter_min(B, ab, loss)                                          # :197
    i1, i2 = an[ia], index[ia]                                                # :200-201
    rows = np.arange(B)
    t1, s1, t2, s2 = tgt[rows, i1], src[rows, i1], tgt[rows, i2], src[rows, i2]
    scale = (t2 - t1) / np.where(s2 != s1, s2 - s1, F32(1e-7))                # :206
    shift = t1 - scale * s1                                                   # :207
    return scale.astype(F32).reshape(bshape), shift.astype(F32).reshape(bshape)


def align_points_scale(points_src, points_tgt, weight):
    """alignment.py:228-243.  (..., n, 3) -> scale (...)"""
    points_src, points_tgt, weight = _f(points_src), _f(points_tgt), _f(weight)
    w3 = np.broadcast_to(weight[..., None], points_src.shape)
    flat = points_src.shape[:-2] + (-1,)
    return align(points_src.reshape(flat), points_tgt.reshape(flat), w3.reshape(flat))[0]


def _points_anchor_solve(points_src, points_tgt, weight, comp_mask):
    points_src, points_tgt, weight = _f(points_src), _f(points_tgt), _f(weight)
    bshape, n = points_src.shape[:-2], points_src.shape[-2]
    src, tgt, w = points_src.reshape(-1, n, 3), points_tgt.reshape(-1, n, 3), weight.reshape(-1, n)
    B = src.shape[0]
    ab, an, _, loss, index = _anchored(src, tgt, w, comp_mask)
    _, ia = scatter_min(B, ab, loss)                                          # :284 / :339
    i2 = index[ia]                                                            # :287 / :341   in [0, 3n)
    i1 = an[ia] * 3 + i2 % 3                                                  # :288 / :342
    m = np.asarray(comp_mask, F32)
    src_a, tgt_a = (src * m).reshape(B, -1), (tgt * m).reshape(B, -1)         # :290-291: the anchor's components (zeros where not anchored)
    rows = np.arange(B)
    t1, s1 = tgt_a[rows, i1], src_a[rows, i1]
    t2, s2 = tgt.reshape(B, -1)[rows, i2], src.reshape(B, -1)[rows, i2]
    scale = (t2 - t1) / np.where(s2 != s1, s2 - s1, F32(1.0))                 # :295 / :348
    shift = (tgt * m)[rows, i1 // 3] - scale[:, None] * (src * m)[rows, i1 // 3]      # :296 / :349
    return scale.astype(F32).reshape(bshape), shift.astype(F32).reshape(*bshape, 3)


def align_points_scale_z_shift(points_src, points_tgt, weight):
    """alignment.py:246-299: shared xyz scale, shift along z only."""
    return _points_anchor_solve(points_src, points_tgt, weight, [0.0, 0.0, 1.0])


def align_points_scale_xyz_shift(points_src, points_tgt, weight):
    """alignment.py:302-354: shared xyz scale, xyz shift."""
    return _points_anchor_solve(points_src, points_tgt, weight, [1.0, 1.0, 1.0])


def align_points_z_shift(points_src, points_tgt, weight):
    """alignment.py:357-376"""
    points_src, points_tgt, weight = _f(points_src), _f(points_tgt), _f(weight)
    s = align(np.ones_like(points_src[..., 2]), points_tgt[..., 2] - points_src[..., 2], weight)[0]
    return np.stack([np.zeros_like(s), np.zeros_like(s), s], axis=-1)


def align_points_xyz_shift(points_src, points_tgt, weight):
    """alignment.py:379-396: one 1-D solve per component."""
    points_src, points_tgt, weight = _f(points_src), _f(points_tgt), _f(weight)
    d = np.swapaxes(points_tgt - points_src, -2, -1)
    return align(np.ones_like(d), d, weight[..., None, :])[0]


def align_affine_lstsq(x, y, w=None):
    """alignment.py:399-415: min sum (sqrt(w) x a + b - sqrt(w) y)^2 (the constant column is NOT weighted in the reference), solved here through
    the normal equations in float64."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    ws = np.ones_like(x) if w is None else np.sqrt(np.asarray(w, np.float64))
    u, v = ws * x, ws * y
    n = x.shape[-1]
    suu, su, suv, sv = (u * u).sum(-1), u.sum(-1), (u * v).sum(-1), v.sum(-1)
    det = suu * n - su * su
    a = (suv * n - su * sv) / det
    b = (suu * sv - su * suv) / det
    return a.astype(F32), b.astype(F32)
