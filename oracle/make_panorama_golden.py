"""TEST INFRASTRUCTURE (not product): pins moge_amd/panorama.py to the reference's moge/utils/panorama.py.

Runs the UNMODIFIED /root/reference/moge/utils/panorama.py (`merge_panorama_depth` :109-191, `split_panorama_image` :39-50,
`get_panorama_cameras` :19-23) in this container and writes its outputs to tests/golden/panorama_ref.npz; tests/test_panorama_reference.py
compares moge_amd.panorama with that file everywhere (the GPU box has no /root/reference) and, where the reference checkout exists, with the
reference run live.

The reference imports two packages this image does not ship.  They are replaced by the DOCUMENTED stubs below, exactly as
oracle/make_golden.py::install_stubs does for `infer()`:

  cv2      `remap` (INTER_LINEAR / INTER_NEAREST, BORDER_CONSTANT / BORDER_REPLICATE) and `resize` (INTER_LINEAR), written out with cv2's
           conventions: sample coordinates are pixel-centre coordinates (pixel i covers [i - 0.5, i + 0.5]), bilinear weights from
           floor(), nearest = rint(), resize maps destination centre (i + 0.5) * scale - 0.5.  cv2 evaluates bilinear remaps with 5-bit
           fixed-point weights (INTER_BITS = 5); the stub uses exact float weights - the uint8 split can therefore differ from real cv2 by
           +-1 LSB, which is also the tolerance of the test.
  utils3d  `np.uv_map`, `np.uv_to_pixel`, `np.project_cv`, `np.unproject_cv`, `np.create_icosahedron_mesh`, `np.intrinsics_from_fov`,
           `np.extrinsics_look_at` with the semantics their call sites need (pixel-centre uv in [0, 1]; pixel = uv * size - 0.5; OpenCV
           camera axes; normalised intrinsics).  utils3d is an un-vendored git dependency (pyproject.toml:23) and is not reachable: the same
           stub runs under both sides, so a wrong CONVENTION there moves both together - "parity unpinned" for utils3d stays in the README.
           What this file pins is everything the reference itself writes: the warps, masks, wrapped gradients, Laplacians, the duplicated
           column-0 equations, the sparse system, lsmr's stopping rule and the coarse-to-fine start.

The stubs are built from moge_amd.panorama's private helpers on purpose: the point is to run the reference's code on the same primitives.

    python oracle/make_panorama_golden.py          # rewrites tests/golden/panorama_ref.npz
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden", "panorama_ref.npz")


def install_panorama_stubs():
    """cv2 / utils3d stand-ins for moge/utils/panorama.py (see the module docstring).  Idempotent."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from moge_amd import panorama as P

    cv2 = sys.modules.get("cv2")
    if cv2 is None or not hasattr(cv2, "remap"):
        cv2 = types.ModuleType("cv2")
        cv2.INTER_NEAREST, cv2.INTER_LINEAR = 0, 1
        cv2.BORDER_CONSTANT, cv2.BORDER_REPLICATE = 0, 1

        def remap(src, map1, map2, interpolation, borderMode=0, **_):
            if interpolation == cv2.INTER_NEAREST:
                assert borderMode == cv2.BORDER_REPLICATE
                return P._remap_nearest(src, map1, map2)
            return P._remap_bilinear(src, map1, map2, "replicate" if borderMode == cv2.BORDER_REPLICATE else "constant")

        def resize(src, dsize, *_a, **_k):          # (the reference passes INTER_LINEAR in the `dst` slot, panorama.py:112: cv2's default IS linear)
            return P._resize_bilinear(src, dsize[1], dsize[0])

        cv2.remap, cv2.resize = remap, resize
        sys.modules["cv2"] = cv2

    u = sys.modules.get("utils3d")
    if u is None:
        u = types.ModuleType("utils3d")
        sys.modules["utils3d"] = u
    npm = getattr(u, "np", None)
    if npm is None:
        npm = types.ModuleType("utils3d.np")
        u.np = npm
        sys.modules["utils3d.np"] = npm
    if not hasattr(npm, "project_cv"):
        def uv_map(*size):
            h, w = size[0] if len(size) == 1 else size
            return P._uv_grid(h, w).astype(np.float32)

        def uv_to_pixel(uv, size):
            h, w = size[:2]
            return np.stack([uv[..., 0] * w - 0.5, uv[..., 1] * h - 0.5], axis=-1)

        def project_cv(points, extrinsics=None, intrinsics=None):
            return P._project(points, extrinsics, intrinsics)

        def unproject_cv(uv, depth, extrinsics=None, intrinsics=None):
            return P._view_rays(uv, extrinsics, intrinsics) * np.asarray(depth)[..., None]

        def create_icosahedron_mesh():
            return P._icosahedron_vertices().astype(np.float32), None

        def intrinsics_from_fov(fov_x=None, fov_y=None):
            fx, fy = 0.5 / np.tan(fov_x / 2), 0.5 / np.tan(fov_y / 2)
            return np.array([[fx, 0, 0.5], [0, fy, 0.5], [0, 0, 1]], dtype=np.float32)

        def extrinsics_look_at(eye, look_at, up):
            assert np.allclose(eye, 0)
            return P._look_at_extrinsics(np.asarray(look_at, dtype=np.float64), up)

        npm.uv_map, npm.uv_to_pixel, npm.project_cv, npm.unproject_cv = uv_map, uv_to_pixel, project_cv, unproject_cv
        npm.create_icosahedron_mesh, npm.intrinsics_from_fov, npm.extrinsics_look_at = create_icosahedron_mesh, intrinsics_from_fov, extrinsics_look_at
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference():
    """The reference module, loaded from its file (not through `moge.utils`, whose package import pulls the whole repo)."""
    import importlib.util
    install_panorama_stubs()
    spec = importlib.util.spec_from_file_location("ref_panorama", os.path.join(REFERENCE_ROOT, "moge", "utils", "panorama.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- the seeded cases (shared with the test: inputs are regenerated there, only reference OUTPUTS are stored) ------------------------------
MERGE_CASES = [                      # (name, width, height, view resolution)
    ("m128", 128, 64, 48),
    ("m256", 256, 128, 64),
    ("m512", 512, 256, 96),          # > 256: exercises the coarse-to-fine start (panorama.py:110-112)
]


def room_distance(d, half=(3.0, 2.0, 1.5), centre=(0.4, -0.3, 0.2)):
    half, centre = np.asarray(half), np.asarray(centre)
    with np.errstate(divide="ignore"):
        t = np.where(d > 0, (centre + half) / d, np.where(d < 0, (centre - half) / d, np.inf))
    return t.min(axis=-1)


def merge_inputs(P, res, seed):
    """12 per-view distance maps of a box room, every view in its own scale and with multiplicative noise (so that the views DISAGREE and the
    least-squares weights matter), and masks with holes: a half-view, a disc, random speckle, one view fully masked."""
    E, Ks = P.get_panorama_cameras()
    rng = np.random.default_rng(seed)
    uv = P._uv_grid(res, res)
    dist, masks = [], []
    for i, (e, k) in enumerate(zip(E, Ks)):
        rays = P._view_rays(uv, e, k)
        rays /= np.linalg.norm(rays, axis=-1, keepdims=True)
        d = room_distance(rays) * rng.uniform(0.5, 2.0) * np.exp(rng.normal(0, 0.02, (res, res)))
        m = np.ones((res, res), bool)
        if i == 3:
            m[:, : res // 2] = False
        if i == 5:
            yy, xx = np.mgrid[:res, :res]
            m[(yy - res * 0.4) ** 2 + (xx - res * 0.6) ** 2 < (res * 0.25) ** 2] = False
        if i == 7:
            m &= rng.random((res, res)) > 0.2
        if i == 9:
            m[:] = False
        dist.append(d.astype(np.float32))
        masks.append(m)
    return E, Ks, dist, masks


def split_input(P, H=192, W=384, seed=3):
    d = P.spherical_uv_to_directions(P._uv_grid(H, W))
    rng = np.random.default_rng(seed)
    img = (d * 0.5 + 0.5) * 200 + rng.uniform(0, 55, (H, W, 3))          # smooth + texture
    return np.clip(img, 0, 255).astype(np.uint8)


def reference_outputs(ref, P):
    out = {}
    E, Ks = ref.get_panorama_cameras()
    out["cam_E"], out["cam_K"] = np.asarray(E), np.stack(Ks)
    for name, w, h, res in MERGE_CASES:
        E_, Ks_, dist, masks = merge_inputs(P, res, seed=w)
        depth, mask = ref.merge_panorama_depth(w, h, dist, masks, E_, Ks_)
        out[name + "_depth"], out[name + "_mask"] = depth.astype(np.float32), mask
    img = split_input(P)
    views = ref.split_panorama_image(img, E, Ks, 64)
    out["split_u8"] = np.stack(views)
    out["split_f32"] = np.stack(ref.split_panorama_image(img.astype(np.float32), E[:3], Ks[:3], 32))
    return out


if __name__ == "__main__":
    ref = load_reference()
    from moge_amd import panorama as P
    out = reference_outputs(ref, P)
    np.savez_compressed(GOLDEN, **out)
    print("wrote", os.path.normpath(GOLDEN), {k: (v.shape, str(v.dtype)) for k, v in out.items()})
