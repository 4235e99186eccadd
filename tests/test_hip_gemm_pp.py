"""GPU: the production fp16 throughput GEMM (gemm_pp.hip: gemm_pp128m16_kernel, every epilogue flavour it ships) against a plain
PyTorch fp32 reference of the same op, and bit-for-bit against the latency-regime kernels of gemm.hip.

Dispatch (gemm.hip launch_gemm): a GEMM goes to the ping-pong kernel when N % 256 == 0, K % 64 == 0 and it has at least PP_MIN_TILES
(96) tiles of 256x256; everything smaller runs gemm_glds_kernel / gemm_kernel.  The tests pin the kernel under test explicitly:
  force = "pp"      moge_tune_set("PP_MIN_TILES", 0)   -> the ping-pong kernel for every eligible shape; PP_KERN picks which of the two
                    product kernels (gemm_pp128p_kernel: persistent, the default; gemm_pp128m16_kernel: one tile per workgroup) - each test runs both
  force = "latency" moge_tune_set("GEMM_PP", 0)        -> gemm.hip only
and also run the full BASELINE shapes (M = 32 x 3601 = 115232 = 450 x 256 + 32: a row tail) under the production dispatch.

Reference: inputs rounded to fp16 (what the kernel reads), products / sums in fp32 (torch matmul on the GPU is the checker, never the
product path), epilogue in fp32, result rounded to fp16 where the kernel stores fp16.  Tolerance: 1.5 fp16 ulp of the output's magnitude
(one rounding + the fp32 summation-order difference) -> max |out - ref| <= 1e-3 * max |ref|, mean error 10x below that."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LOG2E_8 = 0.125 * 1.4426950408889634


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests import hip_util
    return hip_util


@pytest.fixture(params=[0, 2], ids=["pp128m16", "pp128p"])
def kern(request):
    """The throughput kernels the library ships: gemm_pp128p_kernel (persistent, the production default), gemm_pp128m16_kernel (one tile per
    workgroup: its fallback); PP_KERN = -1 is the production setting."""
    return request.param


class Force:
    """Pin the GEMM dispatch for the duration of a block, then restore the production defaults."""

    def __init__(self, mode, kern=-1):
        self.mode, self.kern = mode, kern

    def __enter__(self):
        from moge_amd import _lib as L
        if self.mode == "pp":
            L.tune("PP_MIN_TILES", 0)
            L.tune("PP_KERN", self.kern)
            if self.kern >= 2:
                L.tune("PP_GRID", 24)      # persistent kernel: 3 workgroups per XCD, so even the small shapes walk several tiles each
        elif self.mode == "latency":
            L.tune("GEMM_PP", 0)
        return self

    def __exit__(self, *exc):
        from moge_amd import _lib as L
        L.tune("PP_MIN_TILES", 96)
        L.tune("GEMM_PP", 1)
        L.tune("PP_KERN", -1)
        L.tune("PP_GRID", 0)


def h16(t):
    return t.half().float()


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=g, device="cuda") * scale


def close(out, ref, tol=1e-3, what=""):
    out, ref = out.double(), ref.double()
    scale = float(ref.abs().max())
    err = (out - ref).abs()
    assert float(err.max()) <= tol * scale, (what, float(err.max()) / scale)
    assert float(err.mean()) <= 0.1 * tol * scale, (what, "mean", float(err.mean()) / scale)


def acc_ref(A, W):
    return h16(A) @ h16(W).T


# (M, N, K): a row tail (M % 256 = 32 / 4 / 17), K of the ViT-L layers (1024, 4096) and one short K, N of proj / qkv / fc1
SHAPES_SMALL = [(3601, 1024, 1024), (2 * 3601 + 17, 1024, 4096), (1024 + 32, 768, 768), (3601, 512, 64)]
FULL_M = 32 * 3601            # BASELINE configs[2]: 115232 = 450 * 256 + 32


@pytest.mark.parametrize("M,N,K", SHAPES_SMALL)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_store_flavours_forced_pp(H, kern, M, N, K, act):
    """EPK_STORE <1>, EPK_RELU <6>, EPK_GELU <2> (bias + activation, fp16 row stores)."""
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = acc_ref(A, W) + b
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    with Force("pp", kern):
        out = H.gemm_ex(H.TG_STORE, A, W, b, act=act)["out"]
    close(out, h16(ref), what=f"store act{act}")
    with Force("latency"):
        lat = H.gemm_ex(H.TG_STORE, A, W, b, act=act)["out"]
    assert torch.equal(out, lat), f"pp and latency kernels differ on {int((out != lat).sum())} of {out.numel()} outputs"


@pytest.mark.parametrize("M,N,K", SHAPES_SMALL[:3])
def test_resid_flavour_forced_pp(H, kern, M, N, K):
    """EPK_RESID <0>: x += gamma (acc + bias) on the fp32 residual stream, with the LN-fold producer outputs: the fp16 copy of the
    updated row and the (sum, sum of squares) of each of its 32-column groups (GemmArgs::x16 / ln_part)."""
    A, W, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6)
    gamma, x0 = rnd(N, seed=7), rnd(M, N, seed=8, scale=3.0)
    ref = x0 + gamma * (acc_ref(A, W) + b)
    with Force("pp", kern):
        o = H.gemm_ex(H.TG_RESID, A, W, b, xres=x0, gamma=gamma, want_x16=True)
    close(o["xres"], ref, tol=2e-6, what="resid x")                       # fp32 in, fp32 out: only the summation order differs
    assert torch.equal(o["x16"], h16(o["xres"])), "x16 is not the fp16 rounding of the updated residual"
    g = o["xres"].double().reshape(M, N // 32, 32)
    part = torch.stack([g.sum(-1), (g * g).sum(-1)], dim=-1)
    close(o["ln_part"], part, tol=2e-6, what="ln_part")
    with Force("latency"):
        l = H.gemm_ex(H.TG_RESID, A, W, b, xres=x0, gamma=gamma, want_x16=True)
    for k in ("xres", "x16", "ln_part"):
        assert torch.equal(o[k], l[k]), f"{k}: pp and latency kernels differ on {int((o[k] != l[k]).sum())} entries"
    with Force("pp", kern):                                                     # without the LN-fold outputs (last block's fc2)
        o2 = H.gemm_ex(H.TG_RESID, A, W, b, xres=x0, gamma=gamma)
    assert torch.equal(o2["xres"], o["xres"])


@pytest.mark.parametrize("M,N,K", SHAPES_SMALL[:3] + [(517, 384, 1536)])
def test_resid16_flavour_forced_pp(H, kern, M, N, K):
    """EPK_RESID16 <9>: the proj / fc2 epilogue of a `.half()` model - the residual stream itself is fp16 and is updated in place:
    x16 <- fp16(x16 + gamma (acc + bias)) (one rounding), with the (sum, sum of squares) of every 32-column group of the ROUNDED row."""
    A, W, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6)
    gamma, x0 = rnd(N, seed=7), h16(rnd(M, N, seed=8, scale=3.0))
    ref = x0 + gamma * (acc_ref(A, W) + b)
    pp_ok = N % 256 == 0
    with Force("pp", kern):
        o = H.gemm_ex(H.TG_RESID, A, W, b, gamma=gamma, x16_stream=x0)
    # one fp16 rounding of a value whose fp32 sum may differ in the last bits: within 1 fp16 ulp of the fp32 reference's rounding
    err = (o["x16"] - h16(ref)).abs()
    bound = ref.abs() * 2.0 ** -10 + 2e-5            # one fp16 ulp of the result + the fp32 summation-order difference where x and the update cancel
    assert bool((err <= bound).all()), float((err - bound).max())
    assert float((err > 0).float().mean()) < 5e-3, "more than 0.5 % of the entries round differently from the fp32 reference"
    assert torch.equal(o["x16"], h16(o["x16"]))
    g = o["x16"].double().reshape(M, N // 32, 32)
    close(o["ln_part"], torch.stack([g.sum(-1), (g * g).sum(-1)], dim=-1), tol=2e-6, what="ln_part of the rounded stream")
    with Force("latency"):
        l = H.gemm_ex(H.TG_RESID, A, W, b, gamma=gamma, x16_stream=x0)
    for k in ("x16", "ln_part"):
        assert torch.equal(o[k], l[k]), f"{k}: pp and latency kernels differ on {int((o[k] != l[k]).sum())} entries (pp eligible: {pp_ok})"
    with Force("pp", kern):                                                     # without the statistics (last block's fc2)
        o2 = H.gemm_ex(H.TG_RESID, A, W, b, gamma=gamma, x16_stream=x0, want_part=False)
    assert torch.equal(o2["x16"], o["x16"])


def ln_stats(M, seed):
    mean = rnd(M, seed=seed, scale=0.3)
    rstd = rnd(M, seed=seed + 1).abs() * 0.5 + 0.5
    return torch.stack([mean, rstd], dim=-1).contiguous()


@pytest.mark.parametrize("M,N,K", [(3601, 4096, 1024), (2 * 3601 + 17, 3072, 768)])
def test_gelu_ln_flavour_forced_pp(H, kern, M, N, K):
    """EPK_GELU_LN <7>: fc1 as the consumer of a folded LayerNorm: gelu(rstd (acc - mean c) + b')."""
    A, W, b, c = rnd(M, K, seed=9), rnd(N, K, seed=10, scale=K ** -0.5), rnd(N, seed=11), rnd(N, seed=12)
    mr = ln_stats(M, 13)
    ref = F.gelu(mr[:, 1:2] * (acc_ref(A, W) - mr[:, 0:1] * c) + b)
    with Force("pp", kern):
        out = H.gemm_ex(H.TG_STORE, A, W, b, act=2, ln_mr=mr, ln_c=c)["out"]
    close(out, h16(ref), what="gelu_ln")
    with Force("latency"):
        lat = H.gemm_ex(H.TG_STORE, A, W, b, act=2, ln_mr=mr, ln_c=c)["out"]
    assert torch.equal(out, lat), f"pp and latency kernels differ on {int((out != lat).sum())} outputs"


def qkv_ref(y, B, Ntok, nh, qscale):
    y = y.reshape(B, Ntok, 3, nh, 64).permute(2, 0, 3, 1, 4)             # attention.py:72-74
    return y[0] * qscale, y[1], y[2]


@pytest.mark.parametrize("B,Ntok,nh,fold", [(2, 3601, 16, True), (3, 1370, 12, True), (2, 3601, 16, False), (5, 300, 4, True)])
def test_qkv_flavours_forced_pp(H, kern, B, Ntok, nh, fold):
    """EPK_QKV_LN <8> / EPK_QKV <3>: head-major (B, nh, Ntok, 64) rows for q (pre-scaled by log2(e)/8), k and v; a 256-row tile straddles
    batch items (3601 % 256 != 0)."""
    D = nh * 64
    M, N, K = B * Ntok, 3 * D, D
    A, W, b = rnd(M, K, seed=14), rnd(N, K, seed=15, scale=K ** -0.5), rnd(N, seed=16)
    kw, y = {}, acc_ref(A, W)
    if fold:
        c, mr = rnd(N, seed=17), ln_stats(M, 18)
        kw = dict(ln_mr=mr, ln_c=c)
        y = mr[:, 1:2] * (y - mr[:, 0:1] * c)
    rq, rk, rv = qkv_ref(y + b, B, Ntok, nh, LOG2E_8)
    with Force("pp", kern):
        o = H.gemm_ex(H.TG_QKV, A, W, b, nh=nh, Ntok=Ntok, qscale=LOG2E_8, **kw)
    for k, r in (("q", rq), ("k", rk), ("v", rv)):
        close(o[k], h16(r), what=k)
    with Force("latency"):
        l = H.gemm_ex(H.TG_QKV, A, W, b, nh=nh, Ntok=Ntok, qscale=LOG2E_8, **kw)
    for k in "qkv":
        assert torch.equal(o[k], l[k]), f"{k}: pp and latency kernels differ on {int((o[k] != l[k]).sum())} entries"


@pytest.mark.parametrize("B,pixH,pixW,Cin,Cout", [(2, 60, 60, 1024, 256), (1, 42, 85, 256, 128), (3, 37, 40, 512, 64)])
def test_convt_flavour_forced_pp(H, kern, B, pixH, pixW, Cin, Cout):
    """EPK_CONVT <4>: ConvTranspose2d(k2, s2) as a GEMM to 4*Cout columns + pixel-shuffle store (modules.py:162)."""
    M, N, K = B * pixH * pixW, 4 * Cout, Cin
    A, W, b = rnd(M, K, seed=19), rnd(N, K, seed=20, scale=K ** -0.5), rnd(Cout, seed=21).repeat(4)     # bias replicated per (dy, dx), as model.hip packs it
    y = (acc_ref(A, W) + b).reshape(B, pixH, pixW, 2, 2, Cout).permute(0, 1, 3, 2, 4, 5).reshape(B * 2 * pixH * 2 * pixW, Cout)
    with Force("pp", kern):
        out = H.gemm_ex(H.TG_CONVT, A, W, b, pix=(pixW, pixH), Cout=Cout)["out"].reshape(-1, Cout)
    close(out, h16(y), what="convt")
    with Force("latency"):
        lat = H.gemm_ex(H.TG_CONVT, A, W, b, pix=(pixW, pixH), Cout=Cout)["out"].reshape(-1, Cout)
    assert torch.equal(out, lat)
    # the same through torch's own transposed convolution (layout check independent of the reshape above)
    wt = h16(W).reshape(2, 2, Cout, Cin).permute(3, 2, 0, 1).contiguous()                  # [ci][co][dy][dx]
    ref2 = F.conv_transpose2d(h16(A).reshape(B, pixH, pixW, Cin).permute(0, 3, 1, 2), wt, b[:Cout], stride=2).permute(0, 2, 3, 1)
    close(out, h16(ref2.reshape(-1, Cout)), what="convt vs conv_transpose2d")


@pytest.mark.parametrize("B,pixH,pixW,Cin,Cout", [(2, 60, 60, 1024, 1024), (1, 85, 42, 512, 512)])
def test_uv_flavour_forced_pp(H, kern, B, pixH, pixW, Cin, Cout):
    """EPK_UV <5>: the neck's level-0 1x1 input block with the (u, v) view-plane channels folded into a rank-2 epilogue term
    (v2.py:154-160, modules.py:245): out = acc + bias + wu u(x) + wv v(y), u / v = torch.linspace over the pixel centres."""
    M, N, K = B * pixH * pixW, Cout, Cin
    A, W, b = rnd(M, K, seed=22), rnd(N, K, seed=23, scale=K ** -0.5), rnd(N, seed=24)
    wu, wv = rnd(N, seed=25), rnd(N, seed=26)
    aspect = pixW / pixH
    sx, sy = aspect / (1 + aspect ** 2) ** 0.5, 1 / (1 + aspect ** 2) ** 0.5
    u0, u1, v0, v1 = -sx * (pixW - 1) / pixW, sx * (pixW - 1) / pixW, -sy * (pixH - 1) / pixH, sy * (pixH - 1) / pixH
    u = torch.linspace(u0, u1, pixW, device="cuda")
    v = torch.linspace(v0, v1, pixH, device="cuda")
    uu = u[None, None, :].expand(B, pixH, pixW).reshape(M, 1)
    vv = v[None, :, None].expand(B, pixH, pixW).reshape(M, 1)
    ref = acc_ref(A, W) + b + wu * uu + wv * vv
    with Force("pp", kern):
        out = H.gemm_ex(H.TG_STORE, A, W, b, uv=(wu, wv, u0, u1, v0, v1), pix=(pixW, pixH))["out"]
    close(out, h16(ref), what="uv")
    with Force("latency"):
        lat = H.gemm_ex(H.TG_STORE, A, W, b, uv=(wu, wv, u0, u1, v0, v1), pix=(pixW, pixH))["out"]
    assert torch.equal(out, lat)


def test_full_baseline_shapes_production_dispatch(H):
    """BASELINE configs[2] (moge-2-vitl, batch 32, 3601 tokens): the four block GEMMs and the summed output projection at their real
    sizes under the PRODUCTION dispatch (no tuning switch) - every one of them must land on the ping-pong kernel's result.
    M = 115232 = 450 x 256 + 32 (row tail); out-proj M = 115200, K = 4096."""
    M, D = FULL_M, 1024
    B, Ntok, nh = 32, 3601, 16
    # qkv (QKV_LN <8>)
    A, W, b, c, mr = rnd(M, D, seed=30), rnd(3 * D, D, seed=31, scale=D ** -0.5), rnd(3 * D, seed=32), rnd(3 * D, seed=33), ln_stats(M, 34)
    y = mr[:, 1:2] * (acc_ref(A, W) - mr[:, 0:1] * c) + b
    o = H.gemm_ex(H.TG_QKV, A, W, b, nh=nh, Ntok=Ntok, qscale=LOG2E_8, ln_mr=mr, ln_c=c)
    for k, r in zip("qkv", qkv_ref(y, B, Ntok, nh, LOG2E_8)):
        close(o[k], h16(r), what="full " + k)
    del o, y
    # proj (RESID <0>, K = 1024) and fc2 (RESID <0>, K = 4096), with the LN-fold producer outputs
    for K, seed in ((D, 40), (4 * D, 50)):
        A, W, b = rnd(M, K, seed=seed), rnd(D, K, seed=seed + 1, scale=K ** -0.5), rnd(D, seed=seed + 2)
        gamma, x0 = rnd(D, seed=seed + 3), rnd(M, D, seed=seed + 4, scale=3.0)
        ref = x0 + gamma * (acc_ref(A, W) + b)
        o = H.gemm_ex(H.TG_RESID, A, W, b, xres=x0, gamma=gamma, want_x16=True)
        close(o["xres"], ref, tol=2e-6, what=f"full resid K={K}")
        assert torch.equal(o["x16"], h16(o["xres"]))
        g = o["xres"].double().reshape(M, D // 32, 32)
        close(o["ln_part"], torch.stack([g.sum(-1), (g * g).sum(-1)], dim=-1), tol=2e-6, what="full ln_part")
        del o, ref, g
    # proj / fc2 of a `.half()` model (RESID16 <9>): fp16 stream in place
    for K, seed in ((D, 45), (4 * D, 55)):
        A, W, b = rnd(M, K, seed=seed), rnd(D, K, seed=seed + 1, scale=K ** -0.5), rnd(D, seed=seed + 2)
        gamma, x0 = rnd(D, seed=seed + 3), h16(rnd(M, D, seed=seed + 4, scale=3.0))
        ref = x0 + gamma * (acc_ref(A, W) + b)
        o = H.gemm_ex(H.TG_RESID, A, W, b, gamma=gamma, x16_stream=x0)
        err = (o["x16"] - h16(ref)).abs()
        assert bool((err <= ref.abs() * 2.0 ** -10 + 2e-5).all()), (K, float((err - ref.abs() * 2.0 ** -10).max()))
        g = o["x16"].double().reshape(M, D // 32, 32)
        close(o["ln_part"], torch.stack([g.sum(-1), (g * g).sum(-1)], dim=-1), tol=2e-6, what="full ln_part (fp16 stream)")
        del o, ref, g, err
    # fc1 (GELU_LN <7>)
    A, W, b, c, mr = rnd(M, D, seed=60), rnd(4 * D, D, seed=61, scale=D ** -0.5), rnd(4 * D, seed=62), rnd(4 * D, seed=63), ln_stats(M, 64)
    ref = F.gelu(mr[:, 1:2] * (acc_ref(A, W) - mr[:, 0:1] * c) + b)
    out = H.gemm_ex(H.TG_STORE, A, W, b, act=2, ln_mr=mr, ln_c=c)["out"]
    close(out, h16(ref), what="full fc1")
    del out, ref
    # summed output projections (STORE <1>): M = 32 * 3600, K = 4 * 1024
    Mo = 32 * 3600
    A, W, b = rnd(Mo, 4 * D, seed=70), rnd(D, 4 * D, seed=71, scale=(4 * D) ** -0.5), rnd(D, seed=72)
    out = H.gemm_ex(H.TG_STORE, A, W, b)["out"]
    close(out, h16(acc_ref(A, W) + b), what="full out-proj")


def test_fast_gelu_matches_erf_gelu(H, kern):
    """common.h gelu_fast (sigmoid of a fitted odd polynomial; fp16 outputs only) against the exact-erf GELU the reference uses
    (nn.GELU() default, vision_transformer.py:61), through the ping-pong epilogue, over the whole fp16-relevant range: A = [x_hi, x_lo, 0..]
    and W = [1, 1, 0..] make acc = x_hi + x_lo exactly (both halves are fp16 numbers), the bias adds a per-column offset."""
    M, N, K = 512, 256, 64
    A = torch.zeros(M, K, device="cuda")
    W = torch.zeros(N, K, device="cuda")
    row = torch.linspace(-12.0, 12.0, M, device="cuda")
    hi = row.half().float()
    lo = (row - hi).half().float()
    A[:, 0], A[:, 1] = hi, lo
    W[:, 0], W[:, 1] = 1.0, 1.0
    b = torch.linspace(-0.5, 0.5, N, device="cuda")
    ref = F.gelu(((hi + lo)[:, None] + b[None, :]).double()).float()
    with Force("pp", kern):
        out = H.gemm_ex(H.TG_STORE, A, W, b, act=2)["out"]
    err = (out - ref).abs()
    bound = 1e-5 + ref.abs() * 2.0 ** -11 * 1.01            # 3.7e-6 (fit) + one fp16 rounding of the result
    assert bool((err <= bound).all()), float((err - bound).max())


@pytest.mark.parametrize("M,N,K", [(3601, 1024, 64), (3601, 1024, 128), (700, 768, 192), (3601, 1024, 1024), (2 * 3601, 1024, 4096), (517, 384, 1536)])
def test_latency_kernel_ring_depths_are_bit_identical(H, M, N, K):
    """gemm_glds_kernel 64 x 128 (batch <= 2): the 3-slab LDS ring the library ships (DMA two slabs ahead, counted vmcnt), the double buffer of
    round 2 and a 4-slab ring run the same MFMAs in the same order - same bits, including K of 1, 2 and 3 slabs (the ring's prologue / tail
    cases) and the RESID epilogue; and the 3-slab form against torch."""
    from moge_amd import _lib as L
    A, W, b = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=K ** -0.5), rnd(N, seed=23)
    gamma, x0 = rnd(N, seed=24), rnd(M, N, seed=25, scale=3.0)
    outs = {}
    try:
        for ns in (2, 3, 4):
            L.tune("GLDS_SMALL_NS", ns)
            with Force("latency"):
                outs[ns] = (H.gemm_ex(H.TG_STORE, A, W, b, act=2)["out"], H.gemm_ex(H.TG_RESID, A, W, b, xres=x0, gamma=gamma, want_x16=True))
    finally:
        L.tune("GLDS_SMALL_NS", 3)
    close(outs[3][0], h16(F.gelu(acc_ref(A, W) + b)), what="gelu store")
    close(outs[3][1]["xres"], x0 + gamma * (acc_ref(A, W) + b), tol=2e-6, what="resid x")
    for ns in (2, 4):
        assert torch.equal(outs[ns][0], outs[3][0]), f"ring depth {ns} != 3 (store)"
        for k in ("xres", "x16", "ln_part"):
            assert torch.equal(outs[ns][1][k], outs[3][1][k]), f"ring depth {ns} != 3 ({k})"
