"""Panorama split / merge (moge_amd/panorama.py; reference moge/utils/panorama.py + scripts/infer_panorama.py) against an analytic scene.

cv2 and utils3d are not installed, so the reference's own functions cannot run here ("parity unpinned", see the module header): these tests
pin the pipeline to geometry instead - a box room whose distance along every ray is known in closed form, seen through the 12 cameras, each
view scaled by its own factor as an affine-invariant predictor would return it."""
import numpy as np
import pytest
import torch

from moge_amd import panorama as P


def room_distance(d: np.ndarray, half=(3.0, 2.0, 1.5), centre=(0.4, -0.3, 0.2)) -> np.ndarray:
    """Distance from the origin to the walls of an axis-aligned box (origin inside) along unit directions d (..., 3)."""
    half, centre = np.asarray(half), np.asarray(centre)
    with np.errstate(divide="ignore"):
        t = np.where(d > 0, (centre + half) / d, np.where(d < 0, (centre - half) / d, np.inf))
    return t.min(axis=-1)


def view_distance_maps(E, Ks, res, scales):
    uv = P._uv_grid(res, res)
    out = []
    for e, k, s in zip(E, Ks, scales):
        rays = P._view_rays(uv, e, k)
        rays /= np.linalg.norm(rays, axis=-1, keepdims=True)
        out.append((room_distance(rays) * s).astype(np.float32))
    return out


def test_cameras_are_twelve_right_handed_90_degree_views():
    E, Ks = P.get_panorama_cameras()
    assert E.shape == (12, 4, 4) and len(Ks) == 12
    V = P._icosahedron_vertices()
    for e, v in zip(E, V):
        R = e[:3, :3].astype(np.float64)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and np.linalg.det(R) > 0.999
        assert np.allclose(R[2], v, atol=1e-6)                  # the camera's z axis is the view direction
        assert R[1, 2] < 0                                      # y points down (world up = +z)
        assert np.allclose(e[:3, 3], 0)
    assert np.allclose(P.intrinsics_to_fov_x_deg(np.array(Ks)), 90.0, atol=1e-4)
    # neighbouring view directions of an icosahedron are 63.4 degrees apart: 90-degree views overlap and cover the sphere
    d = P.spherical_uv_to_directions(P._uv_grid(64, 128))
    covered = np.zeros(d.shape[:2], bool)
    for e, k in zip(E, Ks):
        uv, z = P._project(d, e, k)
        covered |= (z > 0) & (uv > 0).all(-1) & (uv < 1).all(-1)
    assert covered.all()


def test_spherical_uv_round_trip_and_conventions():
    uv = P._uv_grid(32, 64)
    d = P.spherical_uv_to_directions(uv)
    assert np.allclose(np.linalg.norm(d, axis=-1), 1.0)
    assert np.allclose(P.directions_to_spherical_uv(d * 3.7), uv, atol=1e-9)
    assert np.allclose(P.spherical_uv_to_directions(np.array([0.5, 0.0])), [0, 0, 1], atol=1e-12)          # v = 0: +z
    assert np.allclose(P.spherical_uv_to_directions(np.array([0.5, 0.5])), [-1, 0, 0], atol=1e-12)         # the image centre looks along -x
    assert np.allclose(P.spherical_uv_to_directions(np.array([0.25, 0.5])), [0, -1, 0], atol=1e-12)


def test_split_samples_the_panorama_along_each_views_rays():
    H, W, res = 256, 512, 64
    d = P.spherical_uv_to_directions(P._uv_grid(H, W))
    pano = np.clip((d * 0.5 + 0.5) * 255, 0, 255).astype(np.uint8)               # colour = direction
    E, Ks = P.get_panorama_cameras()
    views = P.split_panorama_image(pano, E, Ks, res)
    assert len(views) == 12 and views[0].shape == (res, res, 3) and views[0].dtype == np.uint8
    uv = P._uv_grid(res, res)
    for e, k, v in zip(E, Ks, views):
        rays = P._view_rays(uv, e, k)
        rays /= np.linalg.norm(rays, axis=-1, keepdims=True)
        want = (rays * 0.5 + 0.5) * 255
        suv = P.directions_to_spherical_uv(rays)
        away = (suv[..., 0] * W > 1.0) & (suv[..., 0] * W < W - 1.0) & (suv[..., 1] * H > 1.0) & (suv[..., 1] * H < H - 1.0)     # not at the seam / poles
        assert np.abs(v.astype(np.float64) - want)[away].max() < 4.0             # bilinear on a smooth image + uint8 rounding
    f = P.split_panorama_image(pano.astype(np.float32), E[:1], Ks[:1], 8)[0]
    assert f.dtype == np.float32


@pytest.mark.parametrize("width,height", [(256, 128), (512, 256)])
def test_merge_recovers_the_room_up_to_one_global_scale(width, height):
    E, Ks = P.get_panorama_cameras()
    rng = np.random.default_rng(0)
    scales = rng.uniform(0.5, 2.0, 12)                        # every view in its own scale, as an affine-invariant model returns it
    dist = view_distance_maps(E, Ks, 96, scales)
    masks = [np.ones((96, 96), bool) for _ in range(12)]
    merged, mask = P.merge_panorama_depth(width, height, dist, masks, E, Ks)
    assert merged.shape == (height, width) and merged.dtype == np.float32 and mask.all()
    truth = room_distance(P.spherical_uv_to_directions(P._uv_grid(height, width)))
    err = np.log(merged) - np.log(truth)
    err -= np.median(err)
    assert np.abs(err).max() < 0.08 and np.abs(err).mean() < 0.03, (np.abs(err).max(), np.abs(err).mean())     # lsmr stops at the reference's atol = btol = 1e-5: a smooth +-2 % residual


def test_merge_ignores_masked_pixels_and_reports_uncovered_ones():
    E, Ks = P.get_panorama_cameras()
    dist = view_distance_maps(E, Ks, 64, np.ones(12))
    masks = [np.ones((64, 64), bool) for _ in range(12)]
    dist[3] = dist[3].copy()
    dist[3][:, :32] *= 3.0                                    # wrong by a factor of 3 where view 3 says "invalid"
    masks[3][:, :32] = False
    merged, mask = P.merge_panorama_depth(256, 128, dist, masks, E, Ks)
    truth = room_distance(P.spherical_uv_to_directions(P._uv_grid(128, 256)))
    err = np.log(merged) - np.log(truth)
    err -= np.median(err)
    # (as in the reference, a pixel next to the mask edge is bilinear in its invalid neighbour - panorama.py:129-130 - so the edge leaks a
    #  little; with the masked half INCLUDED the error would be log 3 = 1.1 over a sixth of the sphere)
    assert np.percentile(np.abs(err), 95) < 0.08 and np.abs(err).mean() < 0.03, (np.percentile(np.abs(err), 95), np.abs(err).mean())
    masks[3][:, :32] = True
    bad, _ = P.merge_panorama_depth(256, 128, dist, masks, E, Ks)
    err_bad = np.log(bad) - np.log(truth)
    err_bad -= np.median(err_bad)
    assert np.abs(err_bad).mean() > 3 * np.abs(err).mean(), (np.abs(err_bad).mean(), np.abs(err).mean())
    assert mask.mean() > 0.9                                  # the other views cover most of what view 3 gave up
    _, mask2 = P.merge_panorama_depth(256, 128, dist[:1], masks[:1], E[:1], Ks[:1])
    assert 0.05 < mask2.mean() < 0.5                          # one 90-degree view covers a sixth of the sphere


class _RoomModel:
    """Stands in for MoGeModel.infer(): returns the room's point map for each view - in a scale of its own, as the real model does."""
    device = torch.device("cpu")

    def __init__(self, E, Ks):
        self.E, self.Ks, self.calls, self.seen = E, Ks, 0, 0

    def infer(self, image, fov_x=None, apply_mask=True, **kw):
        assert image.dim() == 4 and image.shape[1] == 3 and image.dtype == torch.float32 and float(image.max()) <= 1.0
        assert torch.allclose(fov_x, torch.full_like(fov_x, 90.0), atol=1e-3) and apply_mask is False
        B, _, H, W = image.shape
        uv = P._uv_grid(H, W)
        pts = []
        for b in range(B):
            i = self.seen + b
            rays = P._view_rays(uv, np.eye(4), self.Ks[i])                     # camera frame
            unit = rays / np.linalg.norm(rays, axis=-1, keepdims=True)
            world = unit @ self.E[i][:3, :3].astype(np.float64)
            pts.append(unit * room_distance(world)[..., None] * (0.7 + 0.1 * i))
        self.seen += B
        self.calls += 1
        return {"points": torch.tensor(np.stack(pts), dtype=torch.float32), "mask": torch.ones(B, H, W, dtype=torch.bool)}


def test_infer_panorama_pipeline_end_to_end_with_a_stub_model():
    H, W = 160, 320
    d = P.spherical_uv_to_directions(P._uv_grid(H, W))
    pano = np.clip((d * 0.5 + 0.5) * 255, 0, 255).astype(np.uint8)
    E, Ks = P.get_panorama_cameras()
    model = _RoomModel(E, Ks)
    out = P.infer_panorama(model, pano, resolution=64, batch_size=5, merge_size=(256, 128))
    assert model.calls == 3 and model.seen == 12              # 5 + 5 + 2 views
    assert out["distance"].shape == (H, W) and out["mask"].shape == (H, W) and out["points"].shape == (H, W, 3)
    assert out["mask"].all() and len(out["views"]) == 12
    truth = room_distance(d)
    err = np.log(out["distance"]) - np.log(truth)
    err -= np.median(err)
    assert np.abs(err).max() < 0.1 and np.abs(err).mean() < 0.015
    unit = out["points"] / np.linalg.norm(out["points"], axis=-1, keepdims=True)
    assert np.allclose(unit, d, atol=1e-5)                    # points = distance x the pixel's direction


def test_resize_helpers_follow_cv2_conventions():
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.allclose(P._resize_bilinear(a, 3, 4), a)
    up = P._resize_bilinear(a, 6, 8)
    assert up.shape == (6, 8) and np.isclose(up[0, 0], a[0, 0]) and np.isclose(up[-1, -1], a[-1, -1])      # edges replicated
    assert np.isclose(up[0, 1], 0.25)                                               # (1 + 0.5) / 2 - 0.5 = 0.25 between columns 0 and 1
    m = np.array([[1, 0], [0, 1]], dtype=np.uint8)
    assert np.array_equal(P._resize_nearest(m, 4, 4), np.kron(m, np.ones((2, 2), np.uint8)))


def test_cli_is_importable_without_a_gpu_and_lists_the_reference_flags():
    from click.testing import CliRunner
    from moge_amd.scripts.infer_panorama import main
    r = CliRunner().invoke(main, ["--help"])
    assert r.exit_code == 0
    for flag in ("--input", "--output", "--pretrained", "--device", "--resize", "--resolution_level", "--threshold", "--batch_size", "--splitted",
                 "--maps", "--glb", "--ply"):
        assert flag in r.output, flag


def test_cli_file_conventions_default_is_consistent_and_reference_compat_is_opt_in(tmp_path, monkeypatch):
    """the panorama CLI writes the physically consistent files by default (points.exr R, G, B = x, y, z; GLB texture upright) and the
    reference panorama script's two byte conventions (infer_panorama.py:132,147: no RGB -> BGR conversion, unflipped uvs) only with --reference_compat."""
    import inspect
    import json
    import struct
    from moge_amd.scripts import infer as S, infer_panorama as SP
    src_p, src_i = inspect.getsource(SP), inspect.getsource(S)
    assert 'points[..., ::-1] if reference_compat else points' in src_p
    assert 'save_exr(save_path / "points.exr", out["points"][j])' in src_i           # R, G, B = x, y, z
    assert "vertex_uvs * [1, -1] + [0, 1]" in src_i and "vertex_uvs if reference_compat else vertex_uvs * [1, -1] + [0, 1]" in src_p
    assert "resolution_level=resolution_level" not in src_p                          # accepted, not forwarded (infer_panorama.py:101)
    from click.testing import CliRunner
    assert "--reference_compat" in CliRunner().invoke(SP.main, ["--help"]).output
    # the writers themselves: channel R of the EXR is array[..., 0] ...
    from moge_amd import io as IO
    pts = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3)
    IO.save_exr(tmp_path / "p.exr", pts)
    back = IO.read_exr(tmp_path / "p.exr")                                            # (H, W, 3) in R, G, B order
    assert np.array_equal(back, pts)
    # ... and the GLB texture orientation: a mesh built from a map whose TOP row is red must sample red at its top vertices.  glTF's uv origin is the
    # image's top-left corner, so the stored v of a top-row vertex must be < 0.5 when the caller passes v-up uvs (what the CLI now does)
    H, W = 4, 6
    img = np.zeros((H, W, 3), np.uint8)
    img[0] = (255, 0, 0)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    points = np.stack([xx, -yy, np.ones_like(xx)], -1)                                # y up: the top image row has the largest y
    faces, vertices, colors, uvs = IO.build_mesh_from_map(points, img.astype(np.float32) / 255, IO.uv_map(H, W), tri=True)
    IO.save_glb(tmp_path / "m.glb", vertices, faces, uvs * [1, -1] + [0, 1], img)
    raw = (tmp_path / "m.glb").read_bytes()
    jlen = struct.unpack_from("<I", raw, 12)[0]
    gltf = json.loads(raw[20:20 + jlen])
    bin0 = 20 + jlen + 8
    acc = gltf["accessors"][gltf["meshes"][0]["primitives"][0]["attributes"]["TEXCOORD_0"]]
    bv = gltf["bufferViews"][acc["bufferView"]]
    st = np.frombuffer(raw, "<f4", acc["count"] * 2, bin0 + bv["byteOffset"]).reshape(-1, 2)
    top = vertices[:, 1] == vertices[:, 1].max()
    assert top.any() and (st[top, 1] < 0.5).all() and (st[~top, 1] > st[top, 1].max()).all()
    assert (colors[top][:, 0] > 0.99).all()                                           # and those vertices are the red ones
