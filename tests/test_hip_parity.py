"""GPU: the parity tests proper.  The HIP path (through the Python mirror -> C ABI) against the CPU oracle on the same
seeded inputs, against the committed golden fixtures of the REAL reference, and - at the BASELINE sizes - through
size-independent properties.

Tolerances (north_star): FP32 mode: points / depth / normal / intrinsics within 1e-3 relative, validity mask bit-exact.
FP16 mode is judged against the fp32 oracle with the band the reference's own fp16 path shows vs its fp32 path
(~1.5e-3 abs on O(1) values for ViT-S) - 3e-2 relative here, mask mismatches <= 0.5 %."""
import os

import numpy as np
import pytest
import torch

from tests.golden_util import CASE_BY_NAME, load_case, rel_err, subsample

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-3


@pytest.fixture(scope="module")
def MoGeModel():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moge_amd.model import import_model_class_by_version
    return import_model_class_by_version("v2")


_models = {}


def get_model(MoGeModel, cfg_name, seed, sane, tmp_path_factory):
    from oracle import moge_oracle as O
    key = (cfg_name, seed, sane)
    if key not in _models:
        cfg = O.named_configs()[cfg_name]
        sd = O.synth_state_dict(cfg, seed, sane)
        path = os.path.join(str(tmp_path_factory.mktemp("ckpt")), "model.pt")
        O.save_checkpoint(path, cfg, sd)
        _models[key] = (MoGeModel.from_pretrained(path).to("cuda").eval(), cfg, sd)      # through the reference's loader contract
    return _models[key]


def compare(out, ref, tol, mask_frac=0.0, ill=False):
    assert set(out.keys()) == set(ref.keys())
    for k in ref:
        a = out[k].cpu().numpy() if torch.is_tensor(out[k]) else out[k]
        b = ref[k].cpu().numpy() if torch.is_tensor(ref[k]) else ref[k]
        if b.dtype == np.bool_:
            bad = int((a != b).sum())
            assert bad <= mask_frac * b.size, f"mask: {bad}/{b.size} pixels differ"
        elif mask_frac > 0 or ill:
            fin = np.isfinite(a) & np.isfinite(b)       # fp16 mode: a few mask flips move inf entries
            assert fin.mean() > 0.99 * np.isfinite(b).mean()
            e = np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1.0)
            assert np.quantile(e, 0.999) <= tol, (k, float(np.quantile(e, 0.999)))
        else:
            assert rel_err(a, b) <= tol, (k, rel_err(a, b))


@pytest.mark.parametrize("name", [n for n in CASE_BY_NAME])
def test_fp32_mode_matches_reference_golden_and_oracle(MoGeModel, name, tmp_path_factory):
    from oracle import moge_oracle as O
    case, cfg, sd, x, gold, meta = load_case(name)
    model, _, _ = get_model(MoGeModel, case["config"], case["seed"], case["sane"], tmp_path_factory)
    kw = dict(case["kwargs"]); kw["use_fp16"] = False
    out = model.float().infer(x, **kw)
    ill = not case["sane"]
    st = case.get("stride", 1)
    # (1) committed golden vectors of the real reference
    g = {k[6:]: v for k, v in gold.items() if k.startswith("infer.")}
    o = {k: subsample(k, v.cpu().numpy(), st) for k, v in out.items()}
    compare(o, g, 5e-2 if ill else FP32_TOL, ill=ill)
    # (2) the oracle, live, full resolution
    if name != "vits_house518":
        ref = O.infer(cfg, sd, x, **{k: v for k, v in kw.items() if k != "use_fp16"})
        compare(out, ref, 5e-2 if ill else FP32_TOL, ill=ill)


@pytest.mark.parametrize("name", ["tiny_b2_up", "tiny_b1_down_3d", "tiny_fov_nomask_noproj"])
def test_fp16_mode_within_reference_fp16_band(MoGeModel, name, tmp_path_factory):
    case, cfg, sd, x, gold, meta = load_case(name)
    model, _, _ = get_model(MoGeModel, case["config"], case["seed"], case["sane"], tmp_path_factory)
    kw = dict(case["kwargs"]); kw["use_fp16"] = True
    out = model.float().infer(x, **kw)             # fp32 weights + use_fp16 (autocast analogue)
    out_h = model.half().infer(x, **kw)            # .half() weights
    model.float()
    g = {k[6:]: v for k, v in gold.items() if k.startswith("infer.")}
    for o in (out, out_h):
        compare({k: v.cpu().numpy() for k, v in o.items()}, g, 3e-2, mask_frac=5e-3)


def test_fp16_mode_vits_real_image_vs_reference_golden(MoGeModel, tmp_path_factory):
    """ViT-S decoder dims (256/128/64/32) on the 518x518 example image: the only golden case whose shapes reach every fp16
    throughput kernel (gemm_pp 128/256-wide tiles, attention_pp, conv_pp 64/128-wide with 1..4 Cin chunks, pixel-shuffle
    resampler).  Judged against the REAL reference's fp32 output with the fp16 band."""
    case, cfg, sd, x, gold, meta = load_case("vits_house518")
    model, _, _ = get_model(MoGeModel, case["config"], case["seed"], case["sane"], tmp_path_factory)
    kw = dict(case["kwargs"]); kw["use_fp16"] = True
    st = case.get("stride", 1)
    g = {k[6:]: v for k, v in gold.items() if k.startswith("infer.")}
    try:
        out_h = model.half().infer(x, **kw)
    finally:
        model.float()
    compare({k: subsample(k, v.cpu().numpy(), st) for k, v in out_h.items()}, g, 3e-2, mask_frac=5e-3)


def test_stage_taps_match_oracle(MoGeModel, tmp_path_factory):
    """Stage boundaries of one forward (fp32 mode): LayerNorm'ed ViT taps, cls token, encoder features, every neck level."""
    from oracle import moge_oracle as O
    case, cfg, sd, x, gold, meta = load_case("tiny_b2_up")
    model, _, _ = get_model(MoGeModel, case["config"], case["seed"], case["sane"], tmp_path_factory)
    model.float()
    fwd = model.forward(x, case["kwargs"]["num_tokens"])
    tr = {}
    ref = O.forward(cfg, sd, x, case["kwargs"]["num_tokens"], tr)
    B = x.shape[0]
    taps = torch.cat([t[:, 1:] for t in tr["taps"]], dim=-1).reshape(-1)
    assert rel_err(model.debug_tap("tapcat").cpu().numpy(), taps.numpy()) < 1e-4
    assert rel_err(model.debug_tap("cls").cpu().numpy(), tr["cls"].reshape(-1).numpy()) < 1e-4
    assert rel_err(model.debug_tap("features").cpu().numpy(), tr["features"].permute(0, 2, 3, 1).reshape(-1).numpy()) < 1e-4
    for l, n in enumerate(tr["neck"]):
        assert rel_err(model.debug_tap(f"neck{l}").cpu().numpy(), n.permute(0, 2, 3, 1).reshape(-1).numpy()) < 2e-4, l
    for k in ref:
        assert rel_err(fwd[k].cpu().numpy(), ref[k].numpy()) < 5e-4, k


def test_error_behaviour_matches_reference(MoGeModel, tmp_path_factory):
    """scipy raises ValueError('Residuals are not finite in the initial point.') when the point map overflows."""
    from oracle import moge_oracle as O
    cfg = O.named_configs()["tiny-vits-normal"]
    sd = O.synth_state_dict(cfg, 0, True)
    sd["points_head.output_blocks.4.bias"][2] = 200.0        # exp(200) = inf in fp32
    path = os.path.join(str(tmp_path_factory.mktemp("ckpt")), "bad.pt")
    O.save_checkpoint(path, cfg, sd)
    model = MoGeModel.from_pretrained(path).to("cuda").eval()
    with pytest.raises(ValueError):
        model.infer(torch.rand(1, 3, 64, 64), num_tokens=64, use_fp16=False)


def test_properties_at_baseline_size(MoGeModel, tmp_path_factory):
    """BASELINE config sizes (518x518, T=3600, fp16, vitb) - size-independent properties: batch items are independent
    (sharding contract of SURVEY 8(e)), outputs are deterministic, re-projection is consistent with depth+intrinsics,
    masked pixels are inf / 0, normals are unit length."""
    model, cfg, sd = get_model(MoGeModel, "moge-2-vitb-normal", 0, True, tmp_path_factory)
    model.half()
    g = torch.Generator().manual_seed(0)
    x = torch.rand(3, 3, 518, 518, generator=g)
    out = model.infer(x)
    out2 = model.infer(x)
    for k in out:
        assert torch.equal(out[k], out2[k]), f"{k} not deterministic"
    single = model.infer(x[1])
    for k in out:
        a, b = out[k][1], single[k]
        if a.dtype == torch.bool:
            assert torch.equal(a, b)
        else:
            fin = torch.isfinite(a)
            assert torch.equal(fin, torch.isfinite(b)) and torch.equal(a[fin], b[fin]), f"{k}: batch item depends on its batch"
    m, d, p, K, n = out["mask"], out["depth"], out["points"], out["intrinsics"], out["normal"]
    assert torch.isinf(d[~m]).all() and torch.isinf(p[~m]).all() and (n[~m] == 0).all()
    assert (d[m] > 0).all() and torch.equal(p[..., 2][m], d[m])
    u = (torch.arange(518, device="cuda") + 0.5) / 518
    xx = (u[None, None, :] - 0.5) / K[:, 0, 0, None, None] * d
    assert torch.allclose(xx[m], p[..., 0][m], rtol=1e-5, atol=1e-6)
    assert torch.allclose(n[m].norm(dim=-1), torch.ones_like(d[m]), atol=1e-4)
    model.float()


def test_batch_split_streams_are_bit_identical(MoGeModel, tmp_path_factory):
    """The production path runs a batch >= 8 as two half batches on two internal streams (model.hip forward_dispatch);
    every image is computed independently of its batch, so the split result must equal the single-stream result."""
    from moge_amd import _lib as L
    model, cfg, sd = get_model(MoGeModel, "tiny-vits-normal", 0, True, tmp_path_factory)
    x = torch.rand(9, 3, 84, 112, generator=torch.Generator().manual_seed(3))
    try:
        for half in (False, True):
            if half:
                model.half()
            L.tune("BATCH_SPLIT", 0)
            ref = model.infer(x, num_tokens=108)
            L.tune("BATCH_SPLIT", 1)
            out = model.infer(x, num_tokens=108)
            for k in ref:
                a, b = out[k], ref[k]
                if a.dtype == torch.bool:
                    assert torch.equal(a, b), k
                else:
                    fin = torch.isfinite(b)
                    assert torch.equal(fin, torch.isfinite(a)) and torch.equal(a[fin], b[fin]), f"{k}: split != single stream"
    finally:
        L.tune("BATCH_SPLIT", 1)
        model.float()


def test_eval_plugin_runs_through_the_click_loader(tmp_path_factory):
    """the reference's harness does `Baseline.load.main(args, standalone_mode=False)` then
    `infer_for_evaluation(image, intrinsics)` (moge/scripts/eval_baseline.py:40-42,65-71)."""
    import importlib.util
    from oracle import moge_oracle as O
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg = O.named_configs()["tiny-vits-normal"]
    sd = O.synth_state_dict(cfg, 0, True)
    path = os.path.join(str(tmp_path_factory.mktemp("ckpt")), "model.pt")
    O.save_checkpoint(path, cfg, sd)
    plug = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "baselines", "moge_mi355x.py")
    spec = importlib.util.spec_from_file_location("moge_mi355x_plugin", plug)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    base = mod.Baseline.load.main(["--pretrained", path, "--num_tokens", "108", "--device", "cuda:0"], standalone_mode=False)
    x = torch.rand(3, 84, 112, generator=torch.Generator().manual_seed(5)).cuda()
    K = torch.tensor([[0.9, 0.0, 0.5], [0.0, 1.2, 0.5], [0.0, 0.0, 1.0]], device="cuda")
    out = base.infer_for_evaluation(x, K)
    assert set(out) >= {"points_metric", "depth_metric", "intrinsics"}
    ref = O.infer(cfg, sd, x.cpu(), num_tokens=108, fov_x=float(mod._fov_x_degrees(K.cpu())), apply_mask=False)
    assert rel_err(out["depth_metric"].cpu().numpy(), ref["depth"].numpy()) < FP32_TOL
    assert rel_err(out["intrinsics"].cpu().numpy(), ref["intrinsics"].numpy()) < FP32_TOL


def test_master_blob_cache_is_bit_identical_to_the_checkpoint_load(MoGeModel, tmp_path_factory):
    """save_blob -> from_blob and the from_pretrained sidecar reproduce the .pt-loaded model bit for bit (fp32 and fp16 mode)."""
    from oracle import moge_oracle as O
    cfg = O.named_configs()["tiny-vits-normal"]
    sd = O.synth_state_dict(cfg, 3, True)
    d = str(tmp_path_factory.mktemp("blob"))
    path = os.path.join(d, "model.pt")
    O.save_checkpoint(path, cfg, sd)
    m0 = MoGeModel.from_pretrained(path).to("cuda").eval()
    x = torch.rand(2, 3, 84, 112, generator=torch.Generator().manual_seed(11)).cuda()
    ref32 = m0.infer(x, num_tokens=108, use_fp16=False)
    ref16 = m0.infer(x, num_tokens=108, use_fp16=True)
    blob = os.path.join(d, "model.blob")
    m0.save_blob(blob)
    header, off = MoGeModel.read_blob_header(blob)
    assert header["model_config"]["encoder"]["backbone"] == cfg["encoder"]["backbone"] and off % 4096 == 0
    m1 = MoGeModel.from_blob(blob).to("cuda").eval()
    assert m0.cache_blob() == path + ".mi355x-blob"
    m2 = MoGeModel.from_pretrained(path)                       # now served by the sidecar: no state dict on the host
    assert m2._state is None and m2._blob_path == path + ".mi355x-blob"
    m2 = m2.to("cuda").eval()
    for m in (m1, m2):
        o32 = m.infer(x, num_tokens=108, use_fp16=False)
        o16 = m.infer(x, num_tokens=108, use_fp16=True)
        for k in ref32:
            assert torch.equal(o32[k], ref32[k]), k
            assert torch.equal(o16[k], ref16[k]), k
    # a stale sidecar (checkpoint rewritten later) is ignored
    os.utime(path + ".mi355x-blob", (1, 1))
    assert MoGeModel.from_pretrained(path)._state is not None


@pytest.mark.parametrize("shape,tokens", [((37, 211), 96), ((300, 100), 147), ((64, 64), 16), ((97, 131), 300), ((518, 130), 120)])
def test_odd_shapes_and_token_grids_match_the_oracle_fp32(MoGeModel, tmp_path_factory, shape, tokens):
    """Ragged sizes either side of the resampler: down- AND up-sampling in the same image (37x211 -> 5x30 grid), a 1:3 portrait strip,
    the smallest grids, a prime-sized image, a 4:1 strip of the BASELINE height; token grid = Python half-to-even rounding (v2.py:147)."""
    from oracle import moge_oracle as O
    model, cfg, sd = get_model(MoGeModel, "tiny-vits-normal", 0, True, tmp_path_factory)
    H, W = shape
    x = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(H * 1000 + W))
    out = model.infer(x, num_tokens=tokens, use_fp16=False)
    ref = O.infer(cfg, sd, x, num_tokens=tokens)
    assert out["mask"].shape == (2, H, W) and out["points"].shape == (2, H, W, 3)
    compare(out, ref, FP32_TOL)


@pytest.mark.parametrize("kind", ["black", "white", "flat_gray"])
def test_fp16_mode_on_constant_images_stays_in_band(MoGeModel, tmp_path_factory, kind):
    """Degenerate inputs for the folded LayerNorm (fp16 path: row statistics from partial sums, raw residual in fp16): a constant image gives
    every patch the same embedding, so the token rows differ by the position embedding only.  Outputs must stay finite and within the fp16 band
    of the fp32 oracle, like any other image."""
    from oracle import moge_oracle as O
    model, cfg, sd = get_model(MoGeModel, "tiny-vits-normal", 0, True, tmp_path_factory)
    val = {"black": 0.0, "white": 1.0, "flat_gray": 0.5}[kind]
    x = torch.full((2, 3, 84, 112), val)
    model.half()
    try:
        out = model.infer(x, num_tokens=108, apply_mask=False)
    finally:
        model.float()
    ref = O.infer(cfg, sd, x, num_tokens=108, apply_mask=False)
    for k in ("points", "depth", "intrinsics"):
        a = out[k].float().cpu().numpy()
        assert np.isfinite(a).all(), k
    compare({k: out[k] for k in ref}, ref, 3e-2, mask_frac=0.02)
