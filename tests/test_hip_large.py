"""GPU: inputs and batches far above the fixture sizes.

The reference CLI feeds images at their native size by default (scripts/infer.py:93-101; the shipped examples are 1500 x 1000), where the
resize-back / remap kernels, `finalize_kernel`, `recover_kernel`'s 64 x 64 sub-sampling and `moge_depth_edge_mask` see 3-20x more pixels than any
golden fixture; and nothing in the fixtures runs a batch above 32.  Here:
  * `infer()` of the tiny model at 1000 x 1500 and 3000 x 4000 (B = 2, fp32 mode) against the CPU oracle, every pixel (check_fp32), plus the
    depth-edge mask at that size against the host restatement;
  * the bench model (.half()) at B = 64 and at B = 150 - the second with the batch split off, so the N = 4096 GEMM (M x 4096 x 2 B = 4.4 GB) leaves
    the persistent kernel's 32-bit output offsets (gemm_pp.hip pp_persistent_ok) - items at both ends equal their single-image results bit for bit;
  * the same crossing at GEMM level against torch fp32: every epilogue family at M = 150 x 3601.
Stated behaviour above the 4 GiB line: the dispatch falls back to the one-tile-per-workgroup kernel (64-bit addressing), same arithmetic bit for bit."""
import numpy as np
import pytest
import torch

from tests.golden_util import check_fp32, load_case
from tests.test_hip_parity import MoGeModel, _same, get_model      # noqa: F401  (fixture re-export)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,tokens", [(1000, 1500, 1200), (3000, 4000, 1800)])
def test_native_size_images_match_the_oracle_fp32(MoGeModel, tmp_path_factory, H, W, tokens):
    """1.5 MP (the example images' native size) and 12 MP, batch 2, fp32 mode vs the oracle: every pixel within 1e-3, mask bit-exact except knife-edge
    pixels (see below: identified from the oracle's own margins, at most 2 per million)."""
    from oracle import moge_oracle as O
    model, cfg, sd = get_model(MoGeModel, "tiny-vits-normal", 0, True, tmp_path_factory)
    x = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(H + W))
    out = model.infer(x, num_tokens=tokens, use_fp16=False)
    trace = {}
    ref = O.infer(cfg, sd, x, num_tokens=tokens, trace=trace)
    assert out["points"].shape == (2, H, W, 3) and out["mask"].shape == (2, H, W)
    # mask = sigmoid > 0.5 and z + shift > 0 (v2.py:253,268): at 3-24 M pixels a handful land within the fp32 forward noise (5e-6 relative) of a threshold.
    # ONLY pixels whose mask actually differs are exempt, each must sit within 2e-5 of a threshold by the ORACLE's own margins, and there may be at most
    # 3 + 2 per million of them
    prob = trace["forward"]["mask"].reshape(2, H, W)
    z = trace["forward"]["points"].reshape(2, H, W, 3)[..., 2] + trace["shift"].reshape(2, 1, 1)
    near = ((prob - 0.5).abs() < 2e-5) | (z.abs() < 2e-5 * z.abs().median())
    res = {k: v.cpu().clone() for k, v in out.items()}
    knife = res["mask"] != ref["mask"]
    assert not bool((knife & ~near).any()), f"{int((knife & ~near).sum())} mask pixels differ away from any threshold"
    assert int(knife.sum()) <= 3 + 2 * (2 * H * W) // 1000000, int(knife.sum())
    for k in ("points", "depth", "mask", "normal"):
        if k in res:
            res[k][knife] = ref[k][knife]
    seen = check_fp32(res, ref)
    print(f"[large fp32] {H}x{W} tokens={tokens}: " + " ".join(f"{k}={v:.2e}" for k, v in seen.items()) + f" knife={int(knife.sum())}")
    # the caller-side clean-up kernel at the same size (scripts/infer.py:127)
    from oracle import caller_side as CS
    got = model.depth_edge_mask(out["depth"], out["mask"], rtol=0.04).cpu().numpy()
    d, m = out["depth"].cpu().numpy(), out["mask"].cpu().numpy()
    for b in range(2):
        assert np.array_equal(got[b], m[b] & ~CS.depth_map_edge(d[b], 0.04)), b


def _vitl(MoGeModel, tmp_path_factory):
    case, cfg, sd, x1, gold, meta = load_case("vitl_518_t3600")
    model, _, _ = get_model(MoGeModel, None, None, None, tmp_path_factory, case=case)
    return model


@pytest.mark.parametrize("B,split", [(64, 2), (150, 0)])
def test_batches_above_32_equal_their_single_image_results(MoGeModel, tmp_path_factory, B, split):
    """B = 64 (two 32-image streams) and B = 150 as ONE stream (M = 540150 rows: fc1's output is 4.4 GB, beyond the persistent GEMM's 32-bit offsets ->
    one-tile-per-workgroup kernel; everything else stays persistent): items 0 / B/2 - 1 / B/2 / B - 1 equal their single-image results bit for bit."""
    from moge_amd import _lib as L
    model = _vitl(MoGeModel, tmp_path_factory)
    x = torch.rand(B, 3, 518, 518, generator=torch.Generator().manual_seed(7))
    try:
        model.half()
        L.tune("BATCH_SPLIT", split)
        batch = model.infer(x)
        L.tune("BATCH_SPLIT", 2)
        for k in ("points", "depth"):
            assert batch[k].shape[0] == B
        L.tune("ATTN_KS", 0)         # the batch-invariant form of the one-image attention (the default splits a single image's key range inside the workgroup: within the band, tests/test_hip_parity.py)
        for i in (0, B // 2 - 1, B // 2, B - 1):
            single = model.infer(x[i])
            for k in single:
                _same(batch[k][i], single[k], f"{k}: item {i} of the batch of {B} differs from its single-image result")
    finally:
        L.tune("BATCH_SPLIT", 2)
        L.tune("ATTN_KS", 1)
        model.float()
        torch.cuda.empty_cache()


def test_gemm_dispatch_across_the_4gib_line_matches_torch_and_the_persistent_kernel():
    """M = 150 x 3601 rows.  fc1 (N = 4096, GELU + folded LN): M x ldc x 2 B = 4.42 GB > 2^32 -> pp_persistent_ok() is false, gemm_pp128m16_kernel runs;
    proj (N = 1024, fp16 residual stream) stays persistent.  Both against torch fp32 on sampled row blocks, and the first 100k rows bit for bit
    against the same GEMM run at M = 100000 (persistent kernel, below the line)."""
    from tests import hip_util as H
    from tests.test_hip_gemm_pp import acc_ref, close, h16, rnd
    import torch.nn.functional as F
    M, K = 150 * 3601, 1024
    A = rnd(M, K, seed=11)
    # ---- fc1: GELU, N = 4096
    N = 4096
    W, b = rnd(N, K, seed=12, scale=K ** -0.5), rnd(N, seed=13)
    out = H.gemm_ex(H.TG_STORE, A, W, b, act=2)["out"]
    assert out.shape == (M, N)
    for r0 in (0, 262144 - 128, 524288 - 100, M - 300):                 # incl. the rows either side of the 2^32-byte offset (row 524288)
        rows = slice(r0, min(M, r0 + 300))
        ref = F.gelu(acc_ref(A[rows], W) + b)
        close(out[rows], h16(ref), what=f"fc1 rows {r0}")
    small = H.gemm_ex(H.TG_STORE, A[:100000], W, b, act=2)["out"]
    assert torch.equal(out[:100000 - 256], small[:100000 - 256]), "one-tile-per-workgroup and persistent kernels differ"
    del out, small
    # ---- proj: fp16 residual stream, N = 1024 (persistent at this M: M x 1024 x 2 B = 1.1 GB)
    N = 1024
    W, b, gamma = rnd(N, K, seed=14, scale=K ** -0.5), rnd(N, seed=15), rnd(N, seed=16)
    x0 = h16(rnd(M, N, seed=17, scale=3.0))
    o = H.gemm_ex(H.TG_RESID, A, W, b, gamma=gamma, x16_stream=x0)
    for r0 in (0, 270075 - 64, M - 300):
        rows = slice(r0, min(M, r0 + 300))
        ref = x0[rows] + gamma * (acc_ref(A[rows], W) + b)
        close(o["x16"][rows], h16(ref), tol=1.5e-3, what=f"proj rows {r0}")
