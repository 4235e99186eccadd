"""Panorama pipeline around the hot path (reference: moge/utils/panorama.py + moge/scripts/infer_panorama.py).

An equirectangular panorama is split into 12 perspective views (90-degree field of view, looking at the vertices of an icosahedron), the
views are pushed through `MoGeModel.infer(views, fov_x=..., apply_mask=False)` in batches - the GPU step, `infer_panorama_views` - and the
per-view DISTANCE maps are merged back into one panorama distance map by a least-squares fit of its log-gradients and log-Laplacian to the
views' (each view is only known up to its own scale, which the logarithm turns into an offset that gradients do not see).

    from moge_amd.panorama import infer_panorama
    out = infer_panorama(model, panorama_rgb_uint8)          # {"distance", "mask", "points"} at the panorama's resolution

Same function names and arguments as `moge/utils/panorama.py` (get_panorama_cameras, spherical_uv_to_directions, directions_to_spherical_uv,
split_panorama_image, merge_panorama_depth).  Split and merge are host-side numpy / scipy work, as in the reference (cv2.remap + scipy lsmr
there); what differs is forced by what this image ships:
  * cv2 is not installed: the bilinear / nearest remaps and the bilinear resize are written out here with cv2's conventions (pixel centres at
    integer coordinates, BORDER_CONSTANT 0 for the image split, BORDER_REPLICATE for the merge, half-pixel-centre resize);
  * utils3d is not installed (un-vendored dependency, pyproject.toml:23): the icosahedron, the look-at extrinsics (OpenCV camera: x right,
    y down, z forward; world up = +z) and the uv / pixel conventions (pixel-centre uv in [0, 1], pixel = uv * size - 0.5) are restated.
    Pinned to the reference's own module all the same: oracle/make_panorama_golden.py runs the unmodified moge/utils/panorama.py with cv2 / utils3d
    stand-ins built from the helpers below and stores its outputs (tests/golden/panorama_ref.npz); tests/test_panorama_reference.py holds this file
    to them - merged log-distance within 5e-5 (observed 1e-5 ... 1e-6), masks equal, the uint8 split within 1 LSB - and re-runs the reference live
    where /root/reference exists.  What stays unpinned is the utils3d CONVENTION itself (the same stand-in serves both sides), as for the other
    utils3d call sites (README); tests/test_panorama_cpu.py additionally checks the pipeline against an analytic scene.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------------------------------------------
# cameras and spherical coordinates
# ---------------------------------------------------------------------------------------------------------------------------------------
def _icosahedron_vertices() -> np.ndarray:
    """The 12 unit vertices (0, +-1, +-phi) and cyclic permutations: none lies on the z axis, so `up = +z` is never degenerate."""
    phi = (1.0 + 5.0 ** 0.5) / 2.0
    v = []
    for a in (-1.0, 1.0):
        for b in (-phi, phi):
            v += [(0.0, a, b), (a, b, 0.0), (b, 0.0, a)]
    v = np.array(v, dtype=np.float64)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _look_at_extrinsics(targets: np.ndarray, up=(0.0, 0.0, 1.0)) -> np.ndarray:
    """World -> camera matrices (N, 4, 4) of cameras at the origin looking at `targets` (OpenCV axes: x right, y down, z forward)."""
    z = targets / np.linalg.norm(targets, axis=-1, keepdims=True)
    x = np.cross(z, np.asarray(up, dtype=np.float64))
    x /= np.linalg.norm(x, axis=-1, keepdims=True)
    y = np.cross(z, x)
    E = np.tile(np.eye(4), (len(targets), 1, 1))
    E[:, 0, :3], E[:, 1, :3], E[:, 2, :3] = x, y, z
    return E


def get_panorama_cameras() -> Tuple[np.ndarray, List[np.ndarray]]:
    """panorama.py:19-23 -> (extrinsics (12, 4, 4) float32, 12 x normalised intrinsics (3, 3) of a 90 x 90 degree view)."""
    K = np.array([[0.5, 0.0, 0.5], [0.0, 0.5, 0.5], [0.0, 0.0, 1.0]], dtype=np.float32)          # f = 0.5 / tan(45 deg), in units of the image size
    E = _look_at_extrinsics(_icosahedron_vertices()).astype(np.float32)
    return E, [K] * len(E)


def spherical_uv_to_directions(uv: np.ndarray) -> np.ndarray:
    """panorama.py:26-29: equirectangular uv (u right, v down, both in [0, 1]) -> unit directions; u = 0 / 1 is the +x meridian, v = 0 is +z."""
    theta, phi = (1.0 - uv[..., 0]) * (2.0 * np.pi), uv[..., 1] * np.pi
    s = np.sin(phi)
    return np.stack([s * np.cos(theta), s * np.sin(theta), np.cos(phi)], axis=-1)


def directions_to_spherical_uv(directions: np.ndarray) -> np.ndarray:
    """panorama.py:32-36: the inverse map (directions need not be normalised)."""
    d = directions / np.linalg.norm(directions, axis=-1, keepdims=True)
    u = 1.0 - (np.arctan2(d[..., 1], d[..., 0]) / (2.0 * np.pi)) % 1.0
    v = np.arccos(np.clip(d[..., 2], -1.0, 1.0)) / np.pi
    return np.stack([u, v], axis=-1)


def _uv_grid(height: int, width: int) -> np.ndarray:
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    v = (np.arange(height, dtype=np.float64) + 0.5) / height
    return np.stack(np.meshgrid(u, v, indexing="xy"), axis=-1)


def _view_rays(uv: np.ndarray, extrinsics: np.ndarray, intrinsics: np.ndarray) -> np.ndarray:
    """World-space ray through normalised image coordinates `uv` at camera depth 1 (what unproject_cv(uv, ones) returns for a camera at the origin)."""
    K, R = np.asarray(intrinsics, dtype=np.float64), np.asarray(extrinsics, dtype=np.float64)[:3, :3]
    cam = np.stack([(uv[..., 0] - K[0, 2]) / K[0, 0], (uv[..., 1] - K[1, 2]) / K[1, 1], np.ones_like(uv[..., 0])], axis=-1)
    return cam @ R                                             # R^T applied to row vectors; the translation is zero


def _project(directions: np.ndarray, extrinsics: np.ndarray, intrinsics: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """World directions -> (normalised image uv, camera depth) (project_cv for a camera at the origin)."""
    K, R = np.asarray(intrinsics, dtype=np.float64), np.asarray(extrinsics, dtype=np.float64)[:3, :3]
    cam = directions @ R.T
    z = cam[..., 2]
    zs = np.where(np.abs(z) > 1e-12, z, 1e-12)
    return np.stack([K[0, 0] * cam[..., 0] / zs + K[0, 2], K[1, 1] * cam[..., 1] / zs + K[1, 2]], axis=-1), z


# ---------------------------------------------------------------------------------------------------------------------------------------
# resampling with cv2's conventions
# ---------------------------------------------------------------------------------------------------------------------------------------
def _remap_bilinear(img: np.ndarray, x: np.ndarray, y: np.ndarray, border: str) -> np.ndarray:
    """img (H, W[, C]) sampled at pixel coordinates (x, y); border 'constant' (zeros outside, cv2.BORDER_CONSTANT) or 'replicate'."""
    H, W = img.shape[:2]
    src = img.astype(np.float32).reshape(H, W, -1)
    x0, y0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    fx, fy = (x - x0).astype(np.float32)[..., None], (y - y0).astype(np.float32)[..., None]

    def tap(yy, xx):
        inside = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return v if border == "replicate" else v * inside[..., None]

    out = (tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx) * (1 - fy) + (tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx) * fy
    out = out.reshape(x.shape + img.shape[2:])
    return np.clip(np.rint(out), 0, 255).astype(np.uint8) if img.dtype == np.uint8 else out.astype(img.dtype if img.dtype.kind == "f" else np.float32)


def _remap_nearest(img: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    H, W = img.shape[:2]
    return img[np.clip(np.rint(y).astype(np.int64), 0, H - 1), np.clip(np.rint(x).astype(np.int64), 0, W - 1)]


def _resize_bilinear(img: np.ndarray, height: int, width: int) -> np.ndarray:
    """cv2.resize(..., INTER_LINEAR): destination pixel centre (i + 0.5) * scale - 0.5 in the source, edges replicated."""
    H, W = img.shape[:2]
    x = (np.arange(width, dtype=np.float64) + 0.5) * (W / width) - 0.5
    y = (np.arange(height, dtype=np.float64) + 0.5) * (H / height) - 0.5
    X, Y = np.meshgrid(x, y, indexing="xy")
    return _remap_bilinear(img, X, Y, "replicate")


def _resize_nearest(img: np.ndarray, height: int, width: int) -> np.ndarray:
    H, W = img.shape[:2]
    xi = np.minimum((np.arange(width) * (W / width)).astype(np.int64), W - 1)
    yi = np.minimum((np.arange(height) * (H / height)).astype(np.int64), H - 1)
    return img[yi][:, xi]


# ---------------------------------------------------------------------------------------------------------------------------------------
# split
# ---------------------------------------------------------------------------------------------------------------------------------------
def split_panorama_image(image: np.ndarray, extrinsics: np.ndarray, intrinsics: Sequence[np.ndarray], resolution: int) -> List[np.ndarray]:
    """panorama.py:39-50: the perspective views (resolution x resolution, the image's dtype) of an equirectangular `image` (H, W, 3)."""
    H, W = image.shape[:2]
    uv = _uv_grid(resolution, resolution)
    views = []
    for E, K in zip(extrinsics, intrinsics):
        suv = directions_to_spherical_uv(_view_rays(uv, E, K))
        views.append(_remap_bilinear(image, suv[..., 0] * W - 0.5, suv[..., 1] * H - 0.5, "constant"))
    return views


# ---------------------------------------------------------------------------------------------------------------------------------------
# merge
# ---------------------------------------------------------------------------------------------------------------------------------------
def _difference_operators(width: int, height: int):
    """Sparse operators on a flattened (height, width) map that wraps around in x: forward differences x[i, j] - x[i, j + 1] (every column,
    the last one against column 0) and x[i, j] - x[i + 1, j] (rows 0 .. height - 2), and the 5-point Laplacian with the top / bottom row
    replicated (panorama.py:53-106 builds the same three matrices index by index)."""
    import scipy.sparse as sp
    Ix, Iy = sp.identity(width, format="csr", dtype=np.float32), sp.identity(height, format="csr", dtype=np.float32)
    shift_x = sp.csr_matrix((np.ones(width, np.float32), (np.arange(width), (np.arange(width) + 1) % width)), shape=(width, width))         # picks column j + 1 (wrapped)
    Dx = sp.kron(Iy, Ix - shift_x, format="csr")
    down = sp.csr_matrix((np.ones(height - 1, np.float32), (np.arange(height - 1), np.arange(1, height))), shape=(height - 1, height))       # picks row i + 1
    keep = sp.csr_matrix((np.ones(height - 1, np.float32), (np.arange(height - 1), np.arange(height - 1))), shape=(height - 1, height))
    Dy = sp.kron(keep - down, Ix, format="csr")
    r = np.arange(height)
    up_row = sp.csr_matrix((np.ones(height, np.float32), (r, np.maximum(r - 1, 0))), shape=(height, height))                                # row i - 1, replicated at the top
    down_row = sp.csr_matrix((np.ones(height, np.float32), (r, np.minimum(r + 1, height - 1))), shape=(height, height))
    lap = sp.kron(up_row + down_row, Ix, format="csr") + sp.kron(Iy, shift_x + shift_x.T - 4.0 * Ix, format="csr")
    return Dx, Dy, lap.tocsr()


def merge_panorama_depth(width: int, height: int, distance_maps: Sequence[np.ndarray], pred_masks: Sequence[np.ndarray], extrinsics: Sequence[np.ndarray],
                         intrinsics: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """panorama.py:109-191 -> (panorama distance (height, width) float32, panorama mask).  Coarse to fine: above 256 pixels the half-size
    solution (resized) is the starting point of the solver."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import lsmr

    init = None
    if max(width, height) > 256:
        coarse, _ = merge_panorama_depth(width // 2, height // 2, distance_maps, pred_masks, extrinsics, intrinsics)
        init = _resize_bilinear(coarse, height, width)

    directions = spherical_uv_to_directions(_uv_grid(height, width))
    n = len(distance_maps)
    gx = np.zeros((n, height, width), np.float32); gy = np.zeros((n, height - 1, width), np.float32); lp = np.zeros((n, height, width), np.float32)
    mx = np.zeros((n, height, width), bool); my = np.zeros((n, height - 1, width), bool); ml = np.zeros((n, height, width), bool)
    seen = np.zeros((height, width), bool)
    for i in range(n):
        vh, vw = distance_maps[i].shape
        puv, pz = _project(directions, extrinsics[i], intrinsics[i])
        inside = (pz > 0) & (puv > 0).all(axis=-1) & (puv < 1).all(axis=-1)
        puv = np.clip(puv, 0.0, 1.0)
        px, py = puv[..., 0] * vw - 0.5, puv[..., 1] * vh - 0.5
        logd = np.where(inside, _remap_bilinear(np.log(distance_maps[i]).astype(np.float32), px, py, "replicate"), 0.0).astype(np.float32)
        m = inside & (_remap_nearest(np.asarray(pred_masks[i]).astype(np.uint8), px, py) > 0)
        seen |= m
        # differences towards the right neighbour (wrapping) and the lower neighbour, valid where both ends are
        right, mright = np.roll(logd, -1, axis=1), np.roll(m, -1, axis=1)
        gx[i], mx[i] = logd - right, m & mright
        gy[i], my[i] = logd[:-1] - logd[1:], m[:-1] & m[1:]
        # 5-point Laplacian (x wraps, the top / bottom row is replicated), valid where all five pixels are
        upv, dnv = np.vstack([logd[:1], logd[:-1]]), np.vstack([logd[1:], logd[-1:]])
        mup, mdn = np.vstack([m[:1], m[:-1]]), np.vstack([m[1:], m[-1:]])
        lp[i] = upv + dnv + np.roll(logd, 1, axis=1) + right - 4.0 * logd
        ml[i] = m & mup & mdn & np.roll(m, 1, axis=1) & mright

    def mean_over_views(vals, masks):
        return (vals * masks).sum(axis=0) / np.maximum(masks.sum(axis=0), 1e-3)

    bx, by, bl = mean_over_views(gx, mx), mean_over_views(gy, my), mean_over_views(lp, ml)
    rx, ry, rl = mx.any(axis=0).reshape(-1), my.any(axis=0).reshape(-1), ml.any(axis=0).reshape(-1)
    Dx, Dy, Lap = _difference_operators(width, height)
    # (the reference takes its vertical differences on the map padded by one wrapped column, panorama.py:134-135 with :73-76: the equations
    #  of column 0 enter twice; kept, so that the least-squares weights are the reference's)
    col0 = np.arange(height - 1) * width
    ry0 = ry[col0]
    A = sp.vstack([Dx[rx], Dy[ry], Dy[col0[ry0]], Lap[rl]], format="csr").astype(np.float64)          # (float64: lsmr's norm estimates overflow in float32)
    b = np.concatenate([bx.reshape(-1)[rx], by.reshape(-1)[ry], by.reshape(-1)[col0[ry0]], bl.reshape(-1)[rl]]).astype(np.float64)
    x0 = None if init is None else np.log(init).reshape(-1).astype(np.float64)
    x = lsmr(A, b, atol=1e-5, btol=1e-5, x0=x0, show=False)[0]
    return np.exp(x).reshape(height, width).astype(np.float32), seen


# ---------------------------------------------------------------------------------------------------------------------------------------
# the model-side step and the whole pipeline
# ---------------------------------------------------------------------------------------------------------------------------------------
def intrinsics_to_fov_x_deg(intrinsics: np.ndarray) -> np.ndarray:
    """Horizontal field of view in degrees of NORMALISED intrinsics (fx in units of the image width): 2 atan(0.5 / fx) - what
    `np.rad2deg(utils3d.np.intrinsics_to_fov(K))[0]` evaluates to (utils3d is not vendored: restated from the call site, :100)."""
    K = np.asarray(intrinsics, dtype=np.float64).reshape(-1, 3, 3)
    return np.rad2deg(2.0 * np.arctan(0.5 / K[:, 0, 0])).astype(np.float32)


@torch.inference_mode()
def infer_panorama_views(model, splitted_images: Sequence[np.ndarray], splitted_intrinsics: Sequence[np.ndarray], batch_size: int = 4,
                         **infer_kwargs) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """-> (distance maps, masks), one (H, W) array per view, in view order.  `model` is a moge_amd MoGeModel (v1 or v2) on a GPU; the views are
    uint8 (H, W, 3) arrays of one size, as `split_panorama_image` returns them."""
    dev = model.device
    dist: List[np.ndarray] = []
    masks: List[np.ndarray] = []
    for i in range(0, len(splitted_images), batch_size):
        chunk = np.stack(splitted_images[i:i + batch_size])
        image_tensor = torch.tensor(chunk / 255, dtype=torch.float32, device=dev).permute(0, 3, 1, 2)          # infer_panorama.py:99
        fov_x = torch.tensor(intrinsics_to_fov_x_deg(np.array(splitted_intrinsics[i:i + batch_size])), dtype=torch.float32, device=dev)
        out = model.infer(image_tensor, fov_x=fov_x, apply_mask=False, **infer_kwargs)
        dist.extend(list(out["points"].norm(dim=-1).cpu().numpy()))
        masks.extend(list(out["mask"].cpu().numpy()))
    return dist, masks


def infer_panorama(model, image: np.ndarray, resolution: int = 512, batch_size: int = 4, merge_size: Tuple[int, int] = (1920, 960),
                   **infer_kwargs) -> Dict[str, np.ndarray]:
    """infer_panorama.py:86-121 for one equirectangular uint8 image (H, W, 3): split -> batched infer() -> merge at most at `merge_size`
    (width, height) -> resize to the image.  Returns distance (H, W) float32, mask (H, W) bool, points (H, W, 3) = distance x direction,
    and the per-view intermediates under "views" / "view_distance" / "view_mask"."""
    H, W = image.shape[:2]
    E, Ks = get_panorama_cameras()
    views = split_panorama_image(image, E, Ks, resolution)
    view_dist, view_mask = infer_panorama_views(model, views, Ks, batch_size=batch_size, **infer_kwargs)
    mw, mh = min(merge_size[0], W), min(merge_size[1], H)
    dist, mask = merge_panorama_depth(mw, mh, view_dist, view_mask, E, Ks)
    dist = _resize_bilinear(dist, H, W)
    mask = _resize_nearest(mask.astype(np.uint8), H, W) > 0
    points = dist[:, :, None] * spherical_uv_to_directions(_uv_grid(H, W)).astype(np.float32)
    return {"distance": dist, "mask": mask, "points": points, "views": views, "view_distance": view_dist, "view_mask": view_mask}
