"""the caller-side steps either side of infer() - uint8 ingest on the device, the batch pipeline, the depth-edge
clean-up and the PLY writer.  CPU tests cover the host logic and the oracle restatement; `-m gpu` tests compare the HIP path with it."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import caller_side as CS  # noqa: E402


def test_depth_map_edge_restatement_on_a_hand_case():
    d = np.ones((5, 6), dtype=np.float32)
    d[:, 3:] = 2.0                                    # a step: relative jump 1.0 / depth
    e = CS.depth_map_edge(d, rtol=0.4)
    assert e[:, 2].all() and e[:, 3].all()            # both sides of the step (1/1 and 1/2 > 0.4)
    assert not e[:, :2].any() and not e[:, 4:].any()
    d[0, 0] = np.inf                                  # masked-out pixel (apply_mask sets +inf): its neighbours become edges, itself not (inf/inf = nan)
    e = CS.depth_map_edge(d, rtol=0.4)
    assert not e[0, 0] and e[0, 1] and e[1, 0] and e[1, 1]


def test_ingest_uint8_is_the_reference_callers_expression():
    img = np.random.default_rng(0).integers(0, 256, size=(7, 9, 3), dtype=np.uint8)
    t = CS.ingest_uint8(img)
    assert t.shape == (3, 7, 9) and t.dtype == torch.float32
    assert torch.equal(t, torch.from_numpy((img.astype(np.float64) / 255.0).astype(np.float32)).permute(2, 0, 1))


def test_exr_writer_roundtrip_and_header(tmp_path):
    """depth.exr / points.exr (scripts/infer.py:113,115): uncompressed float32 OpenEXR; +inf (masked pixels) survives; channels Y or R,G,B = x,y,z."""
    from moge_amd import io as IO
    rng = np.random.default_rng(0)
    pts = rng.standard_normal((11, 17, 3)).astype(np.float32)
    depth = np.abs(pts[..., 2]) + 0.1
    depth[3, 5] = np.inf
    IO.save_exr(tmp_path / "d.exr", depth)
    IO.save_exr(tmp_path / "p.exr", pts)
    assert np.array_equal(IO.read_exr(tmp_path / "d.exr"), depth)
    assert np.array_equal(IO.read_exr(tmp_path / "p.exr"), pts)
    raw = (tmp_path / "p.exr").read_bytes()
    assert raw[:4] == bytes([0x76, 0x2F, 0x31, 0x01]) and raw[4] == 2                      # magic 20000630, version 2
    assert raw.index(b"B\x00") < raw.index(b"G\x00") < raw.index(b"R\x00")                # channel list sorted by name
    assert len(raw) == raw.index(b"screenWindowWidth") + len(b"screenWindowWidth\x00float\x00") + 4 + 4 + 1 + 8 * 11 + 11 * (8 + 3 * 17 * 4)


def test_image_mesh_and_glb_container(tmp_path):
    """scripts/infer.py:127-151: grid mesh over the cleaned mask (one quad per fully valid 2x2 block, two triangles each, unused vertices
    dropped) and the .glb container: header, JSON chunk, BIN chunk, accessor counts, PBR parameters of moge/utils/io.py:34-36."""
    import json
    import struct
    from moge_amd import io as IO
    H, W = 6, 8
    rng = np.random.default_rng(1)
    pts = rng.standard_normal((H, W, 3)).astype(np.float32)
    img = (rng.random((H, W, 3)) * 255).astype(np.uint8)
    nrm = rng.standard_normal((H, W, 3)).astype(np.float32)
    mask = np.ones((H, W), dtype=bool)
    mask[2, 3] = False                       # kills the 4 quads around it
    mask[:, 7] = False                       # last column: kills the quads touching it, its vertices go unused
    faces, v, c, uv, n = IO.build_mesh_from_map(pts, img.astype(np.float32) / 255, IO.uv_map(H, W), nrm, mask=mask, tri=True)
    quads = (H - 1) * (W - 2) - 4
    assert faces.shape == (2 * quads, 3) and faces.dtype == np.int32
    assert v.shape[0] == c.shape[0] == uv.shape[0] == n.shape[0] == H * (W - 1) - 1 and faces.max() == v.shape[0] - 1
    # every face's three vertices are pixels of one 2x2 block, inside the mask, and the attribute rows are the pixels' values
    flat_idx = np.flatnonzero(np.isin(np.arange(H * W), np.flatnonzero(mask.reshape(-1))))
    assert np.array_equal(v, pts.reshape(-1, 3)[flat_idx]) and np.array_equal(n, nrm.reshape(-1, 3)[flat_idx])
    ys, xs = np.divmod(flat_idx[faces], W)
    assert (np.ptp(ys, axis=1) == 1).all() and (np.ptp(xs, axis=1) == 1).all()
    e1 = np.stack([xs[:, 1] - xs[:, 0], ys[:, 1] - ys[:, 0]], -1)
    e2 = np.stack([xs[:, 2] - xs[:, 0], ys[:, 2] - ys[:, 0]], -1)
    assert (np.sign(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]) == -1).all()            # one consistent winding
    u0 = IO.uv_map(H, W)
    assert u0.shape == (H, W, 2) and abs(u0[0, 0, 0] - 0.5 / W) < 1e-7 and abs(u0[-1, -1, 1] - (H - 0.5) / H) < 1e-7
    # export conventions of the caller, then the container
    IO.save_glb(tmp_path / "m.glb", v * [1, -1, -1], faces, uv * [1, -1] + [0, 1], img, n * [1, -1, -1])
    raw = (tmp_path / "m.glb").read_bytes()
    magic, ver, total = struct.unpack_from("<4sII", raw, 0)
    assert magic == b"glTF" and ver == 2 and total == len(raw)
    jl, jt = struct.unpack_from("<I4s", raw, 12)
    g = json.loads(raw[20:20 + jl])
    bl, bt = struct.unpack_from("<I4s", raw, 20 + jl)
    assert jt == b"JSON" and bt == b"BIN\x00" and 20 + jl + 8 + bl == len(raw) and g["buffers"][0]["byteLength"] == bl
    prim = g["meshes"][0]["primitives"][0]
    acc = g["accessors"]
    assert acc[prim["attributes"]["POSITION"]]["count"] == v.shape[0] and acc[prim["indices"]]["count"] == faces.size
    assert set(prim["attributes"]) == {"POSITION", "NORMAL", "TEXCOORD_0"}
    pbr = g["materials"][0]["pbrMetallicRoughness"]
    assert pbr["metallicFactor"] == 0.5 and pbr["roughnessFactor"] == 1.0 and pbr["baseColorTexture"]["index"] == 0
    binc = raw[20 + jl + 8:]
    pv = g["bufferViews"][acc[prim["attributes"]["POSITION"]]["bufferView"]]
    pos = np.frombuffer(binc, dtype="<f4", count=3 * v.shape[0], offset=pv["byteOffset"]).reshape(-1, 3)
    assert np.allclose(pos, v * [1, -1, -1])
    tv = g["bufferViews"][acc[prim["attributes"]["TEXCOORD_0"]]["bufferView"]]
    tex = np.frombuffer(binc, dtype="<f4", count=2 * v.shape[0], offset=tv["byteOffset"]).reshape(-1, 2)
    assert np.allclose(tex, uv, atol=1e-6)                    # glTF's own v axis points down: the OpenGL flip is undone inside the file
    iv = g["bufferViews"][g["images"][0]["bufferView"]]
    from PIL import Image
    import io as _io
    assert np.array_equal(np.asarray(Image.open(_io.BytesIO(binc[iv["byteOffset"]:iv["byteOffset"] + iv["byteLength"]]))), img)


def test_colorize_helpers():
    from moge_amd import io as IO
    d = np.linspace(0.5, 5.0, 200, dtype=np.float32).reshape(10, 20)
    d[0, 0] = np.inf
    d[9, 0] = -1.0
    col = IO.colorize_depth(d)
    # moge/utils/vis.py:7-18: an infinite depth (a masked pixel of infer(apply_mask=True)) passes `depth > 0`, becomes disparity 0 and is painted
    # the colour map's far end; only depth <= 0 / NaN pixels are black
    assert col.shape == (10, 20, 3) and col.dtype == np.uint8 and (col[0, 0] == [94, 79, 162]).all() and (col[9, 0] == 0).all()
    sky = d.copy(); sky[:4] = np.inf                              # > 0.1 % masked: the reference's min_disp becomes 0 and the whole image shifts
    cs = IO.colorize_depth(sky)
    disp = 1.0 / np.where(sky > 0, sky, np.nan)
    lo, hi = np.nanquantile(disp, 0.001), np.nanquantile(disp, 0.99)
    assert lo == 0.0 and (cs[:4] == [94, 79, 162]).all() and not (cs[5:] == col[5:]).all()
    near, far = col[0, 1].astype(int), col[-1, -1].astype(int)
    assert near[0] > near[2] and far[2] > far[0]                 # Spectral over 1 - disparity: near = red end, far = blue end
    n = np.zeros((2, 2, 3), dtype=np.float32); n[..., 2] = 1.0
    assert (IO.colorize_normal(n)[0, 0] == [127, 127, 0]).all()


def test_save_ply_layout_roundtrip(tmp_path):
    from moge_amd.io import masked_point_cloud, save_ply
    rng = np.random.default_rng(1)
    pts = rng.standard_normal((4, 5, 3)).astype(np.float32)
    nrm = rng.standard_normal((4, 5, 3)).astype(np.float32)
    img = rng.integers(0, 256, size=(4, 5, 3), dtype=np.uint8)
    mask = rng.random((4, 5)) > 0.3
    v, c, n = masked_point_cloud(pts, mask, img, nrm)
    assert v.shape == (int(mask.sum()), 3) and np.allclose(v, pts[mask] * [1, -1, -1]) and np.allclose(c, img[mask] / 255)
    p = tmp_path / "pc.ply"
    save_ply(p, v, None, c, n)
    raw = p.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and f"element vertex {len(v)}" in lines and "element face 0" in lines
    rec = np.frombuffer(body, dtype=[("p", "<f4", (3,)), ("n", "<f4", (3,)), ("c", "u1", (3,))])
    assert len(rec) == len(v) and np.array_equal(rec["p"], v) and np.array_equal(rec["n"], n)
    assert np.array_equal(rec["c"], np.clip(c * 255, 0, 255).astype(np.uint8))
    save_ply(tmp_path / "tri.ply", v[:3], np.array([[0, 1, 2]]))
    assert (tmp_path / "tri.ply").read_bytes().endswith(bytes([3]) + np.array([0, 1, 2], dtype="<i4").tobytes())


# ------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def model(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moge_amd.model import import_model_class_by_version
    from oracle import moge_oracle as O
    cfg = O.named_configs()["tiny-vits-normal"]
    sd = O.synth_state_dict(cfg, 0, True)
    path = os.path.join(str(tmp_path_factory.mktemp("ckpt")), "model.pt")
    O.save_checkpoint(path, cfg, sd)
    return import_model_class_by_version("v2").from_pretrained(path).to("cuda").eval()


@pytest.mark.gpu
@pytest.mark.parametrize("fp16", [False, True])
def test_infer_uint8_equals_infer_of_the_callers_float_tensor(model, fp16):
    rng = np.random.default_rng(3)
    imgs = rng.integers(0, 256, size=(3, 70, 98, 3), dtype=np.uint8)
    (model.half() if fp16 else model.float())
    try:
        x = torch.stack([CS.ingest_uint8(im) for im in imgs]).cuda()            # what scripts/infer.py:98 uploads
        ref = model.infer(x, num_tokens=108, use_fp16=fp16)
        out = model.infer_uint8(torch.from_numpy(imgs), num_tokens=108, use_fp16=fp16)
        for k in ref:
            assert torch.equal(out[k], ref[k]), k
        one = model.infer_uint8(torch.from_numpy(imgs[1]), num_tokens=108, use_fp16=fp16)     # (H, W, 3): batch dim squeezed like infer()
        assert one["depth"].shape == (70, 98)
        with pytest.raises(ValueError):
            model.infer_uint8(torch.zeros(3, 70, 98, dtype=torch.uint8))
    finally:
        model.float()


@pytest.mark.gpu
def test_depth_edge_mask_matches_the_restatement(model):
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, size=(2, 70, 98, 3), dtype=np.uint8)
    out = model.infer_uint8(torch.from_numpy(imgs), num_tokens=108, use_fp16=False)
    depth, mask = out["depth"].cpu().numpy(), out["mask"].cpu().numpy()
    assert np.isinf(depth[~mask]).all()
    for rtol in (0.04, 0.005):
        got = model.depth_edge_mask(out["depth"], out["mask"], rtol=rtol).cpu().numpy()
        assert np.array_equal(got, mask & ~CS.depth_map_edge(depth, rtol))
    # synthetic steps, borders, masked holes, no mask argument, 2-D input
    d = np.abs(rng.standard_normal((61, 47))).astype(np.float32) + 0.5
    d[10:20, 5:9] = np.inf
    got = model.depth_edge_mask(torch.from_numpy(d), None, rtol=0.3).cpu().numpy()
    assert got.shape == d.shape and np.array_equal(got, ~CS.depth_map_edge(d, 0.3))


@pytest.mark.gpu
def test_pipeline_is_bit_identical_to_direct_calls_and_keeps_order(model):
    from moge_amd.pipeline import InferPipeline
    rng = np.random.default_rng(7)
    batches = [rng.integers(0, 256, size=(n, 56, 84, 3), dtype=np.uint8) for n in (3, 3, 3, 2)]      # ragged tail
    pipe = InferPipeline(model, 3, 56, 84, num_tokens=96, use_fp16=True)
    got = list(pipe.run(iter(batches)))
    assert len(got) == len(batches) and model.sync_on_infer
    for b, g in zip(batches, got):
        pad = np.concatenate([b, np.zeros((3 - len(b), 56, 84, 3), np.uint8)]) if len(b) < 3 else b
        ref = model.infer_uint8(torch.from_numpy(pad), num_tokens=96, use_fp16=True)
        for k in ref:
            assert g[k].shape[0] == len(b)
            assert np.array_equal(g[k], ref[k][:len(b)].cpu().numpy(), equal_nan=True), k
    with pytest.raises(ValueError):
        list(pipe.run(iter([np.zeros((1, 10, 10, 3), np.uint8)])))


@pytest.mark.gpu
def test_cli_writes_every_output_the_reference_cli_writes(tmp_path):
    """`python -m moge_amd.scripts.infer` (reference: moge/scripts/infer.py:18-156) on a folder with two image sizes: image.jpg, depth_vis.png,
    depth.exr, points.exr, mask.png, normal.png, fov.json, mesh.glb, pointcloud.ply per image; the EXR depth equals infer()'s depth."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    from PIL import Image
    from click.testing import CliRunner
    from moge_amd import io as IO
    from moge_amd.model import import_model_class_by_version
    from moge_amd.scripts.infer import main as cli
    from oracle import moge_oracle as O
    cfg = O.named_configs()["tiny-vits-normal"]
    ckpt = str(tmp_path / "model.pt")
    O.save_checkpoint(ckpt, cfg, O.synth_state_dict(cfg, 0, True))
    rng = np.random.default_rng(5)
    src = tmp_path / "in"
    (src / "sub").mkdir(parents=True)
    imgs = {"a.png": (84, 112), "b.png": (84, 112), "sub/c.png": (70, 98)}
    for name, (h, w) in imgs.items():
        Image.fromarray((rng.random((h, w, 3)) * 255).astype(np.uint8)).save(src / name)
    out = tmp_path / "out"
    r = CliRunner().invoke(cli, ["-i", str(src), "-o", str(out), "--pretrained", ckpt, "--num_tokens", "108", "--batch", "2"], catch_exceptions=False)
    assert r.exit_code == 0, r.output
    model = import_model_class_by_version("v2").from_pretrained(ckpt).to("cuda").eval()
    for name, (h, w) in imgs.items():
        d = out / name[:-4]
        for f in ("image.jpg", "depth_vis.png", "depth.exr", "points.exr", "mask.png", "normal.png", "fov.json", "mesh.glb", "pointcloud.ply"):
            assert (d / f).exists(), (name, f)
        depth = IO.read_exr(d / "depth.exr")
        pts = IO.read_exr(d / "points.exr")
        assert depth.shape == (h, w) and pts.shape == (h, w, 3)
        u8 = torch.from_numpy(np.asarray(Image.open(src / name).convert("RGB")))
        ref = model.infer_uint8(u8, num_tokens=108, use_fp16=False)
        assert np.array_equal(depth, ref["depth"].cpu().numpy()) and np.array_equal(pts, ref["points"].cpu().numpy())
        fov = json.load(open(d / "fov.json"))
        assert 1.0 < fov["fov_x"] < 179.0 and 1.0 < fov["fov_y"] < 179.0
        assert (d / "mesh.glb").read_bytes()[:4] == b"glTF"


def test_host_depth_map_edge_equals_the_restatement_the_device_kernel_is_tested_against():
    """moge_amd.io.depth_map_edge (host numpy for callers without a model handle, scripts/infer_baseline.py) = oracle/caller_side's restatement of
    utils3d.np.depth_map_edge, incl. infinite (masked) pixels and the image border."""
    from moge_amd.io import depth_map_edge
    rng = np.random.default_rng(0)
    d = rng.random((37, 53)).astype(np.float32) + 0.5
    d[3, 4] = np.inf
    d[10:12, 20] = np.inf
    d[0, :5] = 3.0
    for r in (0.03, 0.2, 0.5):
        assert np.array_equal(depth_map_edge(d, r), CS.depth_map_edge(d, r)), r


def _edge_with_mask_loops(d, rtol, m):
    """depth_map_edge(depth, rtol, mask=mask) by plain loops: only in-image, unmasked neighbours take part; masked pixels are never edges."""
    H, W = d.shape
    out = np.zeros((H, W), bool)
    for y in range(H):
        for x in range(W):
            if not m[y, x]:
                continue
            vals = [d[yy, xx] for yy in range(max(0, y - 1), min(H, y + 2)) for xx in range(max(0, x - 1), min(W, x + 2)) if m[yy, xx]]
            out[y, x] = (np.float32(max(vals)) + np.float32(-min(vals))) / d[y, x] > rtol
    return out


def test_host_depth_map_edge_with_mask_ignores_masked_neighbours():
    """infer_baseline.py:123 calls depth_map_edge(depth, rtol, mask=mask) - a valid pixel beside a masked one is NOT an edge for that reason
    alone (without the argument the +inf neighbour makes it one and the mesh is eroded along every mask border)."""
    from moge_amd.io import depth_map_edge
    rng = np.random.default_rng(1)
    d = (rng.random((29, 41)).astype(np.float32) * 0.02 + 1.0)
    d[:, 20:] += 1.0                                     # one real step edge
    m = np.ones(d.shape, bool)
    m[5:9, 3:12] = False
    m[0, :] = False
    m[14, 19:23] = False
    dm = np.where(m, d, np.inf).astype(np.float32)
    for r in (0.01, 0.03, 0.3):
        got = depth_map_edge(dm, r, mask=m)
        assert np.array_equal(got, _edge_with_mask_loops(d, r, m)), r
        assert not got[~m].any()
    smooth = depth_map_edge(dm, 0.3, mask=m)
    assert not smooth[4, 3:12].any() and not smooth[1, :18].any() and not smooth[1, 22:].any()      # rows next to the holes: no erosion (the real step sits at columns 19 / 20)
    assert depth_map_edge(dm, 0.3)[4, 3:12].all()                          # ... which the mask-less form does erode


@pytest.mark.gpu
def test_device_depth_edge_mask_with_nan_at_masked_pixels_equals_the_mask_aware_host_form(model):
    """what scripts/infer_baseline.py hands the device kernel: NaN at masked pixels = depth_map_edge(..., mask=mask) (fmaxf / fminf skip NaN)"""
    from moge_amd.io import depth_map_edge
    rng = np.random.default_rng(2)
    d = (rng.random((83, 57)).astype(np.float32) * 0.05 + 1.0)
    d[:, 30:] *= 1.5
    m = rng.random(d.shape) > 0.2
    z = np.where(m, d, np.nan).astype(np.float32)
    for r in (0.02, 0.1):
        got = model.depth_edge_mask(torch.from_numpy(z), torch.from_numpy(m), rtol=r).cpu().numpy()
        assert np.array_equal(got, m & ~depth_map_edge(d, r, mask=m)), r


def test_cli_group_lists_the_commands_of_the_reference_group_that_exist_here():
    """moge/scripts/cli.py:14-22: `moge <command>`; here infer, infer_baseline, infer_panorama (same option names as the reference's commands)."""
    from click.testing import CliRunner
    from moge_amd.scripts import cli as C
    from moge_amd.scripts import infer, infer_baseline, infer_panorama
    r = CliRunner().invoke(C.cli, ["--help"])
    assert r.exit_code == 0 and all(n in r.output for n in ("infer", "infer_baseline", "infer_panorama"))
    assert C.cli.get_command(None, "infer") is infer.main and C.cli.get_command(None, "infer_baseline") is infer_baseline.main
    assert C.cli.get_command(None, "infer_panorama") is infer_panorama.main and C.cli.get_command(None, "train") is None
    r = CliRunner().invoke(C.cli, ["infer_baseline", "--help"])
    assert r.exit_code == 0 and "--baseline" in r.output
    opts = {o for p in infer.main.params for o in p.opts}
    assert {"--input", "--fov_x", "--output", "--pretrained", "--version", "--device", "--fp16", "--resize", "--resolution_level", "--num_tokens", "--threshold",
            "--maps", "--glb", "--ply", "--show"} <= opts                                                          # scripts/infer.py:18-33
    opts = {o for p in infer_baseline.main.params for o in p.opts}
    assert opts == {"--baseline", "--input", "-i", "--output", "-o", "--size", "--skip", "--maps", "--ply", "--glb", "--threshold"}      # infer_baseline.py:17-26
    opts = {o for p in infer_panorama.main.params for o in p.opts}
    assert {"--input", "--output", "--pretrained", "--device", "--resize", "--resolution_level", "--threshold", "--batch_size", "--splitted", "--maps", "--glb",
            "--ply", "--show"} <= opts                                                                              # infer_panorama.py:15-28


@pytest.mark.gpu
def test_infer_baseline_cli_drives_the_plugin_by_path(tmp_path):
    """`moge infer_baseline --baseline baselines/moge_mi355x.py ...` (infer_baseline.py:16-137): the plugin file is loaded by path, its own options
    come after the script's, every key it returns is written under the reference's file names; EXR contents equal the plugin's infer()."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib.util
    import json
    from PIL import Image
    from click.testing import CliRunner
    from moge_amd import io as IO
    from moge_amd.scripts.infer_baseline import main as cli
    from oracle import moge_oracle as O
    cfg = O.named_configs()["tiny-vits-normal"]
    ckpt = str(tmp_path / "model.pt")
    O.save_checkpoint(ckpt, cfg, O.synth_state_dict(cfg, 0, True))
    rng = np.random.default_rng(8)
    src = tmp_path / "in"
    (src / "sub").mkdir(parents=True)
    im = (rng.random((84, 112, 3)) * 255).astype(np.uint8)
    Image.fromarray(im).save(src / "sub" / "a.png")
    plug = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "baselines", "moge_mi355x.py")
    out = tmp_path / "out"
    args = ["--baseline", plug, "-i", str(src), "-o", str(out), "--maps", "--ply", "--glb", "--pretrained", ckpt, "--version", "v2", "--num_tokens", "108"]
    r = CliRunner().invoke(cli, args, catch_exceptions=False)
    assert r.exit_code == 0, r.output
    d = out / "sub" / "a"
    for f in ("image.jpg", "points_metric.exr", "depth_metric.exr", "depth_metric_vis.png", "fov.json", "mesh.ply", "mesh.glb"):
        assert (d / f).exists(), f
    spec = importlib.util.spec_from_file_location("moge_mi355x_plugin_cli", plug)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    base = mod.Baseline.load.main(["--pretrained", ckpt, "--version", "v2", "--num_tokens", "108"], standalone_mode=False)
    ref = base.infer(torch.from_numpy(im.astype(np.float32) / 255.0).permute(2, 0, 1).cuda())
    assert np.array_equal(IO.read_exr(d / "depth_metric.exr"), ref["depth_metric"].cpu().numpy())
    assert np.array_equal(IO.read_exr(d / "points_metric.exr"), ref["points_metric"].cpu().numpy())
    fov = json.load(open(d / "fov.json"))
    assert np.allclose(np.array(fov["intrinsics"]), ref["intrinsics"].cpu().numpy()) and 1.0 < fov["fov_x"] < 179.0
    assert (d / "mesh.glb").read_bytes()[:4] == b"glTF"
    mtime = (d / "fov.json").stat().st_mtime_ns
    r = CliRunner().invoke(cli, args + ["--skip"], catch_exceptions=False)                    # --skip: an existing output folder is left alone
    assert r.exit_code == 0 and (d / "fov.json").stat().st_mtime_ns == mtime
