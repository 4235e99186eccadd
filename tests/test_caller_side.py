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
