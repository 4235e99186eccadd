"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the caller-side steps either side of `infer()`
(SURVEY 8(f-2)).  Nothing under moge_amd/ imports this.

* `ingest_uint8`  - moge/scripts/infer.py:98: `torch.tensor(image / 255, dtype=torch.float32, device=device).permute(2, 0, 1)`
  (numpy true-divides uint8 by a Python int in float64; the tensor constructor rounds to float32 once).
* `depth_map_edge` - `utils3d.np.depth_map_edge(depth, rtol=threshold)` as called at moge/scripts/infer.py:127.  utils3d is an
  un-vendored dependency pinned at 3fab839f (pyproject.toml:23) and absent from /root/reference: **parity unpinned**.  Restated
  from its published algorithm: diff = max_pool_2d(depth, 3, 1, 1) + max_pool_2d(-depth, 3, 1, 1) with -inf padding,
  edge = (diff / depth > rtol) under np.errstate(all='ignore').
"""
import numpy as np
import torch


def ingest_uint8(image_hwc_u8: np.ndarray) -> torch.Tensor:
    return torch.tensor(image_hwc_u8 / 255, dtype=torch.float32).permute(2, 0, 1)


def _max_pool_3x3(x: np.ndarray) -> np.ndarray:
    H, W = x.shape[-2:]
    pad = np.full(x.shape[:-2] + (H + 2, W + 2), -np.inf, dtype=x.dtype)
    pad[..., 1:-1, 1:-1] = x
    out = np.full_like(x, -np.inf)
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, pad[..., dy:dy + H, dx:dx + W])
    return out


def depth_map_edge(depth: np.ndarray, rtol: float) -> np.ndarray:
    with np.errstate(all="ignore"):
        diff = _max_pool_3x3(depth) + _max_pool_3x3(-depth)
        return (diff / depth) > rtol
