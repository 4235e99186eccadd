"""Error metrics shared by the golden generator (reference-fp16 drift statistics) and the parity tests.  TEST INFRASTRUCTURE ONLY.

All errors are PER PIXEL and relative to that pixel's own magnitude (north_star: "within 1e-3 relative"), not to a global
scale and not with an absolute floor of 1:

  points      ||a - b||_2 / max(||b||_2, FLOOR_FRAC * median ||b||_2)          per pixel (3-vector)
  depth       |a - b|     / max(|b|,     FLOOR_FRAC * median |b|)              per pixel
  normal      ||a - b||_2                                                      per pixel inside BOTH masks (b is a unit vector; pixels that are
                                                                               0 = masked in exactly one array are mask flips, counted apart)
  intrinsics  |a - b|     / max(|b|, FLOOR_FRAC)                               per entry (entries are 0, 0.5, 1, fx, fy)
  other       |a - b|     / max(|b|,     FLOOR_FRAC * median |b|)

The floor only matters where a point's norm / depth is below 5 % of the scene's median (possible with apply_mask=False, where
z + shift may cross 0): there the error is measured against 5 % of the scene scale instead of dividing by ~0.
Only entries finite in BOTH arrays enter the error; the number of entries whose finiteness differs is returned beside it
(masked pixels are +inf in points / depth, v2.py:285-289: a mask flip moves an entry between the two sets)."""
from __future__ import annotations

import numpy as np

FLOOR_FRAC = 0.05


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def pixel_errors(name: str, a, b):
    """-> (errors over the both-finite entries (1-D float64), number of entries whose finiteness differs, number of entries)"""
    a, b = _np(a).astype(np.float64), _np(b).astype(np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if name in ("points", "normal") and a.shape[-1] == 3:
        fa, fb = np.isfinite(a).all(-1), np.isfinite(b).all(-1)
        both = fa & fb
        with np.errstate(invalid="ignore"):
            d = np.linalg.norm(np.where(both[..., None], a - b, 0.0), axis=-1)[both]
        if name == "normal":
            # outside the validity mask a normal is exactly 0 (v2.py:289): a pixel that is 0 in one array and a unit vector in the other is a
            # MASK FLIP (error 1 by construction) - counted with the pattern mismatches, not in the error distribution; pixels masked in
            # both arrays carry no information and are left out like the +inf pixels of points / depth
            za = (np.where(both[..., None], a, 1.0) == 0).all(-1)
            zb = (np.where(both[..., None], b, 1.0) == 0).all(-1)
            keep = ~za[both] & ~zb[both]
            return d[keep], int((fa != fb).sum() + (za != zb).sum()), int(fa.size)
        nb = np.linalg.norm(np.where(both[..., None], b, 0.0), axis=-1)[both]
        floor = FLOOR_FRAC * (np.median(nb) if nb.size else 1.0)
        return d / np.maximum(nb, max(floor, 1e-30)), int((fa != fb).sum()), int(fa.size)
    fa, fb = np.isfinite(a), np.isfinite(b)
    both = fa & fb
    d = np.abs(a[both] - b[both])
    mb = np.abs(b[both])
    floor = FLOOR_FRAC if name == "intrinsics" else FLOOR_FRAC * (np.median(mb) if mb.size else 1.0)
    return d / np.maximum(mb, max(floor, 1e-30)), int((fa != fb).sum()), int(fa.size)


def summarize(name: str, a, b) -> dict:
    """max / p99.9 of the per-pixel error and the finiteness-mismatch fraction (JSON-serialisable)."""
    e, nmis, n = pixel_errors(name, a, b)
    return dict(max=float(e.max()) if e.size else 0.0, p999=float(np.quantile(e, 0.999)) if e.size else 0.0,
                nonfinite_mismatch=nmis / max(n, 1))


def mask_flips(a, b) -> float:
    a, b = _np(a), _np(b)
    return float((a != b).sum()) / max(a.size, 1)
