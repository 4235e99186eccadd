"""GPU: every HIP kernel class against a plain PyTorch fp32 reference of the same op (torch ops on the GPU are the
checker here, never the product path).  FP32 mode uses the exact-fp32 MFMA: tolerance 1e-4 relative to the output
scale; FP16 mode (fp16 storage, fp32 accumulate): 2e-2."""
import numpy as np
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {0: 2e-4, 1: 2e-2}


def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests import hip_util
    return hip_util


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,N,K,act", [(300, 256, 192, 0), (129, 64, 72, 1), (77, 32, 40, 0), (1000, 1152, 384, 2), (128, 128, 64, 0),
                                       (4097, 384, 1536, 0), (515, 96, 288, 0),
                                       # shapes the fp16 ping-pong kernels take (gemm_pp.hip): 256x256 full-line tiles, K tails of
                                       # the 64-byte-row variant, M tails, GELU / ReLU epilogues
                                       (700, 1024, 1024, 2), (513, 512, 64, 0), (1111, 256, 96, 1), (2049, 768, 3072, 0)])
def test_gemm(H, prec, M, N, K, act):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = A.cuda() @ W.cuda().T + b.cuda()
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    out = H.gemm(prec, A, W, b, act)
    assert relmax(out, ref) < TOL[prec]


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("D", [384, 768, 1024])
def test_layernorm(H, prec, D):
    g = torch.Generator().manual_seed(D)
    x = torch.randn(777, D, generator=g) * 3 + 0.5
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    ref = F.layer_norm(x.cuda(), (D,), w.cuda(), b.cuda(), 1e-6)
    assert relmax(H.layernorm(prec, x, w, b), ref) < (1e-5 if prec == 0 else 2e-3)


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,nh,N", [(1, 1, 64), (2, 3, 130), (1, 2, 200), (1, 6, 1370), (1, 2, 3601)])
def test_attention(H, prec, B, nh, N):
    g = torch.Generator().manual_seed(N)
    q, k, v = (torch.randn(B, nh, N, 64, generator=g) for _ in range(3))
    q = q * 1.5
    v[:, :, N // 2] += 5.0                      # asymmetric values: a transposed / permuted key order cannot pass
    ref = F.scaled_dot_product_attention(q.cuda(), k.cuda(), v.cuda()).permute(0, 2, 1, 3).reshape(B, N, nh * 64)
    assert relmax(H.attention(prec, q, k, v), ref) < (2e-5 if prec == 0 else 1e-2)


@pytest.mark.parametrize("B,nh,N", [(2, 3, 130), (1, 2, 300), (1, 6, 1370), (1, 2, 3601)])
def test_attention_64_queries_per_wave_is_bit_identical(H, B, nh, N):
    """attn_pp16mq_kernel<4> (64 queries per wave: what large grids run) against <2> (32 per wave): the overflow guard decides per 16-query
    block, so a query's result does not depend on its wave-mates - bit-identical outputs, here with late dominant keys that force guard trips
    in some blocks and not in their neighbours; both also against SDPA."""
    from moge_amd import _lib as L
    g = torch.Generator().manual_seed(N + 1)
    q, k, v = (torch.randn(B, nh, N, 64, generator=g) for _ in range(3))
    q = q * 1.5
    k[:, :, N // 2 + 3] = q[:, :, 5] * 6.0          # one query (block 0) sees a huge late score; its neighbours in other blocks do not
    k[:, 0, N - 2] = q[:, 0, min(70, N - 1)] * 5.0
    ref = F.scaled_dot_product_attention(q.cuda(), k.cuda(), v.cuda()).permute(0, 2, 1, 3).reshape(B, N, nh * 64)
    outs = {}
    for kern in (1, 2):
        L.tune("ATTN_KERN", kern)
        try:
            outs[kern] = H.attention(1, q, k, v)
        finally:
            L.tune("ATTN_KERN", 3)
        assert relmax(outs[kern], ref) < 1e-2, kern
    assert torch.equal(outs[1], outs[2])


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "moge_amd", "lib", "obj", ".experiments")),
                    reason="attn_pp16x_kernel (tools/experiments/) is compiled by `python -m moge_amd.build --experiments` only")
@pytest.mark.parametrize("prio", [0, 1])
@pytest.mark.parametrize("B,nh,N", [(2, 3, 130), (1, 2, 512), (1, 2, 513), (1, 2, 600), (2, 2, 900), (1, 8, 1370), (1, 2, 3571), (1, 2, 3601), (1, 1, 64), (1, 1, 65), (3, 1, 1)])
def test_attention_ping_pong_kernel_is_bit_identical(H, B, nh, N, prio):
    """attn_pp16x_kernel (round 5: 8-wave workgroups, the two waves of a SIMD alternate a matrix phase and a softmax phase, 512 queries share one
    K / V stream) forced with ATTN_KERN = 4 against attn_pp16mq_kernel<2>: the same arithmetic per query, so BIT-identical outputs - across
    sequence lengths that exercise every split of the queries (N < 512: all of them on the attn_pp16mq tail launch; 512 / 513 / 600: full
    blocks + a tail launch; 900 / 3571: a partially filled ping-pong workgroup with query-less waves; 3601: 7 blocks + 17 tail queries), the
    single-tile and two-tile rings, and late dominant keys that trip the overflow guard in some query blocks and not in their neighbours."""
    from moge_amd import _lib as L
    g = torch.Generator().manual_seed(N + 17)
    q, k, v = (torch.randn(B, nh, N, 64, generator=g) for _ in range(3))
    q = q * 1.5
    if N > 80:
        k[:, :, N // 2 + 3] = q[:, :, 5] * 6.0
        k[:, 0, N - 2] = q[:, 0, min(70, N - 1)] * 5.0
        if N > 700:
            k[:, :, 650] = q[:, :, 600] * 6.0           # a trip inside the second group of waves (queries 256-511 of a block) / the second block
    ref = F.scaled_dot_product_attention(q.cuda(), k.cuda(), v.cuda()).permute(0, 2, 1, 3).reshape(B, N, nh * 64)
    outs = {}
    for kern in (1, 4):
        L.tune("ATTN_KERN", kern)
        L.tune("ATTN_X_PRIO", prio)
        try:
            outs[kern] = H.attention(1, q, k, v)
        finally:
            L.tune("ATTN_KERN", 3)
            L.tune("ATTN_X_PRIO", 0)
        assert relmax(outs[kern], ref) < 1e-2, kern
    assert torch.equal(outs[1], outs[4])


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "moge_amd", "lib", "obj", ".experiments")),
                    reason="attn_pp16sk_kernel (tools/experiments/) is compiled by `python -m moge_amd.build --experiments` only")
@pytest.mark.parametrize("qb", [2, 4])
@pytest.mark.parametrize("B,nh,N,wgs", [(1, 16, 3601, 0),          # the batch-1 shape of the bench: 8 XCD ranges x 96 workgroups, 34.4 tiles each
                                        (1, 8, 1370, 0), (1, 2, 3601, 0), (1, 2, 3601, 5), (2, 3, 130, 1), (2, 3, 130, 4), (2, 3, 130, 17),
                                        (1, 2, 300, 3), (1, 2, 300, 7), (1, 8, 700, 2), (1, 1, 64, 1), (1, 1, 65, 2), (3, 1, 1, 1), (1, 2, 513, 9)])
def test_attention_stream_k(H, B, nh, N, wgs, qb):
    """attn_pp16sk_kernel (round 6 experiment, measured 5-8 us slower at one image and therefore not in the product library): attn_pp16mq's body over equal contiguous ranges of (query block, key tile)
    units; a query block cut into segments is combined by the last-arriving workgroup in segment order.  Forced (ATTN_SK = 2) with partitions that give
    every case - one workgroup walking several whole query blocks (wgs 1), segments of 1-2 tiles (wgs 17 at two tiles per block), three segments per
    block, a segment that is only the masked last tile, ranges that start / end on block boundaries, B * nh a multiple of 8 (8 ranges) or not (1) - with late
    dominant keys that trip the overflow guard inside some segments.  Checked against SDPA at the fp16 tolerance, against the unsplit kernel within a few fp16
    ulps of the output scale (same fp32 terms, different order), for run-to-run bit-identity (the combine order is fixed) and with a second launch on the
    same workspace (the counters are left zero)."""
    from moge_amd import _lib as L
    g = torch.Generator().manual_seed(N + 31 * wgs)
    q, k, v = (torch.randn(B, nh, N, 64, generator=g) for _ in range(3))
    q = q * 1.5
    v[:, :, N // 2] += 5.0
    if N > 80:
        k[:, :, N // 2 + 3] = q[:, :, 5] * 6.0
        k[:, 0, N - 2] = q[:, 0, min(70, N - 1)] * 5.0
    ref = F.scaled_dot_product_attention(q.cuda(), k.cuda(), v.cuda()).permute(0, 2, 1, 3).reshape(B, N, nh * 64)
    L.tune("ATTN_SK", 0)
    try:
        plain = H.attention(1, q, k, v)
        L.tune("ATTN_SK", 2); L.tune("ATTN_SK_QB", qb); L.tune("ATTN_SK_WGS", wgs); L.tune("ATTN_SK_MIN_TILES", 1 if wgs == 0 and N < 2000 else 12)
        a = H.attention(1, q, k, v)
        b = H.attention(1, q, k, v)
        L.tune("ATTN_SK_TWICE", 1)
        c = H.attention(1, q, k, v)
    finally:
        for key, dflt in (("ATTN_SK", 0), ("ATTN_SK_QB", 2), ("ATTN_SK_WGS", 0), ("ATTN_SK_MIN_TILES", 12), ("ATTN_SK_TWICE", 0)):
            L.tune(key, dflt)
    assert relmax(a, ref) < 1e-2
    assert relmax(a, plain) < 2e-3
    assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("ks", [1, 2, 4])
@pytest.mark.parametrize("B,nh,N", [(1, 16, 3601),           # one image of the bench: 240 blocks of 256 queries, 29 + 28 key tiles (the dispatch's own choice with ks = 1)
                                    (1, 12, 3601), (1, 6, 1370), (1, 16, 1370), (1, 2, 3601), (2, 3, 130), (1, 2, 300), (1, 2, 513), (1, 3, 65), (1, 1, 128),
                                    (3, 1, 129), (1, 8, 700), (2, 2, 897)])
def test_attention_key_split_inside_the_workgroup(H, B, nh, N, ks):
    """attn_pp16ks_kernel<QB> (round 6, batch-1 latency): 8-wave workgroups, waves 0-3 run attn_pp16mq's body over the first half of the key tiles, waves 4-7 over
    the rest, combined through LDS in a fixed order.  Forced (ATTN_KS = 2 / 4: 32 / 64 queries per wave) on every split - even / odd tile counts (the second group's
    extra barrier), two tiles (each half one tile, the second one the masked last tile), query-less waves in both groups, 17-query tails - and through the dispatch's
    own choice (ATTN_KS = 1), with late dominant keys that trip the overflow guard in either half.  Against SDPA at the fp16 tolerance, against the unsplit kernel
    within a few fp16 ulps of the output scale (the same fp32 terms in another order), and run to run bit for bit (no atomics, fixed combine order)."""
    from moge_amd import _lib as L
    g = torch.Generator().manual_seed(N + 7 * ks)
    q, k, v = (torch.randn(B, nh, N, 64, generator=g) for _ in range(3))
    q = q * 1.5
    v[:, :, N // 2] += 5.0
    if N > 80:
        k[:, :, N // 4 + 3] = q[:, :, 5] * 6.0              # first half of the keys
        k[:, 0, N - 2] = q[:, 0, min(70, N - 1)] * 5.0      # second half, the masked last tile
        k[:, :, (3 * N) // 4] = q[:, :, N - 1] * 6.0        # second half, seen by the last query (a tail wave)
    ref = F.scaled_dot_product_attention(q.cuda(), k.cuda(), v.cuda()).permute(0, 2, 1, 3).reshape(B, N, nh * 64)
    try:
        L.tune("ATTN_KS", 0)
        plain = H.attention(1, q, k, v)
        L.tune("ATTN_KS", ks)
        a = H.attention(1, q, k, v)
        b = H.attention(1, q, k, v)
    finally:
        L.tune("ATTN_KS", 1)
    assert relmax(a, ref) < 1e-2
    assert relmax(a, plain) < 2e-3
    assert torch.equal(a, b)
    # the second group starts from the running max the unsplit kernel carries out of tile 0, so its P operands are the unsplit kernel's: without dominant late keys
    # (no overflow guard trips that raise the max in one key range and not in the other) the two forms differ by fp32 summation order only - a handful of last-bit
    # differences after the fp16 rounding (observed 0.03-0.15 % of the values, rms 1e-6)
    q2, k2, v2 = (torch.randn(B, nh, N, 64, generator=g) for _ in range(3))
    try:
        L.tune("ATTN_KS", 0)
        plain2 = H.attention(1, q2, k2, v2)
        L.tune("ATTN_KS", ks)
        a2 = H.attention(1, q2, k2, v2)
    finally:
        L.tune("ATTN_KS", 1)
    assert (a2 != plain2).float().mean().item() < 5e-3
    assert relmax(a2, plain2) < 1e-3


def test_attention_online_softmax_rescale(H):
    """rule 26: force the running-max rescale branch - one key in a LATE tile dominates one query."""
    g = torch.Generator().manual_seed(7)
    B, nh, N = 1, 1, 300
    q, k, v = (torch.randn(B, nh, N, 64, generator=g) for _ in range(3))
    k[0, 0, 250] = q[0, 0, 10] * 4.0            # raw score ~ 4*|q|^2/8 >> others, appears in tile 3
    ref = F.scaled_dot_product_attention(q.double(), k.double(), v.double()).permute(0, 2, 1, 3).reshape(B, N, 64)
    for prec in (0, 1):
        assert relmax(H.attention(prec, q, k, v), ref) < (2e-5 if prec == 0 else 1e-2)


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,Hh,Ww,Cin,Cout,relu", [(2, 13, 17, 32, 32, False), (1, 9, 20, 64, 128, True), (1, 31, 5, 32, 64, True), (1, 8, 8, 128, 256, False),
                                                   # conv_pp.hip (fp16 halo kernel): 64-wide / 128-wide tiles, 1 and several Cin chunks,
                                                   # partial 16x16 tiles on every border
                                                   (2, 20, 33, 64, 64, True), (1, 37, 18, 128, 64, False), (1, 16, 16, 256, 128, True),
                                                   (3, 7, 50, 192, 256, False)])
def test_conv3x3(H, prec, B, Hh, Ww, Cin, Cout, relu):
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, Hh, Ww, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    xin = F.relu(x) if relu else x
    ref = F.conv2d(F.pad(xin.cuda(), (1, 1, 1, 1), mode="replicate"), w.cuda(), b.cuda()).permute(0, 2, 3, 1)
    out = H.conv3x3(prec, x.permute(0, 2, 3, 1), w, b, relu_in=relu)
    assert relmax(out, ref) < TOL[prec]


@pytest.mark.parametrize("grid", [3, 13])
@pytest.mark.parametrize("B,Hh,Ww,Cin,Cout,relu", [(3, 50, 37, 64, 64, True), (2, 40, 70, 128, 128, False), (2, 33, 21, 256, 256, True)])
def test_conv3x3_persistent_walks_tiles(H, grid, B, Hh, Ww, Cin, Cout, relu):
    """conv_pp_kernel is persistent (a workgroup walks a list of tiles, the next tile's halo is requested from the epilogue of the current one).
    A small workgroup cap makes these small problems walk several tiles per workgroup - with 3 workgroups also the 'fewer ranges than XCDs'
    split - and the result must be the one-tile-per-workgroup result bit for bit (cap = number of tiles)."""
    from moge_amd import _lib as L
    g = torch.Generator().manual_seed(Cin + Hh)
    x = torch.randn(B, Hh, Ww, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    outs = []
    for cap in (grid, 1 << 20):
        L.tune("CONV_GRID", cap)
        try:
            outs.append(H.conv3x3(1, x, w, b, relu_in=relu))
        finally:
            L.tune("CONV_GRID", 0)
    assert torch.equal(outs[0], outs[1])
    xin = F.relu(x) if relu else x
    ref = F.conv2d(F.pad(xin.permute(0, 3, 1, 2).cuda(), (1, 1, 1, 1), mode="replicate"), w.cuda(), b.cuda()).permute(0, 2, 3, 1)
    assert relmax(outs[0], ref) < TOL[1]


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("Cin,Cout,Hh,Ww", [(64, 32, 11, 7), (64, 32, 19, 33), (128, 64, 9, 21)])
def test_conv3x3_fused_bilinear_up2(H, prec, Cin, Cout, Hh, Ww):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, Cin, Hh, Ww, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    up = F.interpolate(x.cuda(), scale_factor=2, mode="bilinear", align_corners=False)
    ref = F.conv2d(F.pad(up, (1, 1, 1, 1), mode="replicate"), w.cuda(), b.cuda()).permute(0, 2, 3, 1)
    assert relmax(H.conv3x3(prec, x.permute(0, 2, 3, 1), w, b, up2=True), ref) < TOL[prec]


@pytest.mark.parametrize("prec", [0, 1])
def test_convtranspose2x2(H, prec):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 128, 6, 9, generator=g)
    w = torch.randn(128, 64, 2, 2, generator=g) / 128 ** 0.5
    b = torch.randn(64, generator=g)
    ref = F.conv_transpose2d(x.cuda(), w.cuda(), b.cuda(), stride=2).permute(0, 2, 3, 1)
    assert relmax(H.convt2x2(prec, x.permute(0, 2, 3, 1), w, b), ref) < TOL[prec]


@pytest.mark.parametrize("Hh,Ww,rows,cols", [(98, 126, 10, 12), (140, 150, 7, 8), (518, 518, 60, 60), (300, 500, 9, 16)])
def test_preprocess_matches_antialiased_interpolate(H, Hh, Ww, rows, cols):
    g = torch.Generator().manual_seed(Hh)
    img = torch.rand(2, 3, Hh, Ww, generator=g)
    ref = F.interpolate(img, (rows * 14, cols * 14), mode="bilinear", align_corners=False, antialias=True)     # CPU reference (ATen)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    ref = (ref - mean) / std
    assert float((H.preprocess(img, rows, cols).cpu() - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("rows,cols", [(10, 12), (60, 60), (42, 85), (37, 37), (7, 8)])
def test_posembed_bicubic_kludge(H, rows, cols):
    from oracle import moge_oracle as O
    g = torch.Generator().manual_seed(rows)
    pos = torch.randn(1, 1 + 37 * 37, 384, generator=g)
    ref = O.pos_embed_for_grid(pos, rows, cols)[0]
    assert float((H.posembed(pos[0], rows, cols).cpu() - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("fixed", [False, True])
def test_recover_matches_oracle_lm(H, fixed):
    from oracle import moge_oracle as O
    g = torch.Generator().manual_seed(11)
    B, Hh, Ww = 3, 70, 90
    uv = O.view_plane_uv(Ww, Hh)
    z = 1.0 + 2.0 * torch.rand(B, Hh, Ww, generator=g)
    f_true = torch.tensor([0.7, 1.1, 1.6]).view(B, 1, 1, 1)
    s_true = torch.tensor([0.2, -0.3, 0.05]).view(B, 1, 1)
    xy = uv[None] * (z + s_true)[..., None] / f_true + 0.01 * torch.randn(B, Hh, Ww, 2, generator=g)
    pts = torch.cat([xy, z[..., None]], dim=-1)
    pts[2] = torch.randn(Hh, Ww, 3, generator=g)                 # ill-posed image
    mask = torch.rand(B, Hh, Ww, generator=g) > 0.3
    focal = torch.tensor([0.8, 1.0, 1.3]) if fixed else None
    f_ref, s_ref = O.recover_focal_shift(pts, mask, focal)
    f, s, status = H.recover(pts, mask, focal)
    assert status == 0
    np.testing.assert_allclose(s.cpu().numpy()[:2], s_ref.numpy()[:2], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(f.cpu().numpy()[:2], f_ref.numpy()[:2], rtol=1e-4)
    np.testing.assert_allclose(s.cpu().numpy()[2], s_ref.numpy()[2], rtol=2e-2, atol=1e-4)      # ill-posed: trajectory-sensitive


def test_recover_fallback_and_nonfinite(H):
    pts = torch.rand(2, 70, 70, 3) + 0.5
    mask = torch.zeros(2, 70, 70, dtype=torch.bool)
    mask[1, 0, 0] = True                 # 0 and 1 sampled valid points -> focal=1, shift=0 (geometry_torch.py:153-156)
    f, s, status = H.recover(pts, mask)
    assert status == 0 and f.tolist() == [1.0, 1.0] and s.tolist() == [0.0, 0.0]
    pts[0, :, :, 2] = 0.0                # z + 0 == 0 -> inf residuals at x0 -> scipy raises ValueError
    f, s, status = H.recover(pts, torch.ones(2, 70, 70, dtype=torch.bool))
    assert status == -5


def test_onnx_mode_forward_matches_oracle_raw_outputs(H):
    """docs/onnx.md: the exported graph is the raw forward() (points, normal, mask probability, metric scale) in onnx_compatible_mode."""
    import os
    import tempfile
    from moge_amd.model import import_model_class_by_version
    from oracle import moge_oracle as O
    cfg = O.named_configs()["tiny-vits-normal"]
    sd = O.synth_state_dict(cfg, 0, True)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.pt")
        O.save_checkpoint(path, cfg, sd)
        model = import_model_class_by_version("v2").from_pretrained(path).to("cuda").eval()
    model.onnx_compatible_mode = True
    assert model.onnx_compatible_mode is True
    x = torch.rand(2, 3, 140, 150, generator=torch.Generator().manual_seed(31))
    for tokens in (56, 1369):                                  # down-sampling grid; the native 37x37 grid (not bypassed in this mode)
        fwd = model.forward(x, tokens)
        ref = O.forward(cfg, sd, x, tokens, onnx_compatible_mode=True)
        off = O.forward(cfg, sd, x, tokens)
        for k in ref:
            assert relmax(fwd[k], ref[k]) < 5e-4, (tokens, k, relmax(fwd[k], ref[k]))
        assert relmax(off["points"], ref["points"]) > 1e-3, "the flag must change the result (AA off / pos-embed by size)"


@pytest.mark.parametrize("Hh,Ww,OH,OW", [(98, 126, 153, 197), (140, 150, 87, 93), (518, 518, 700, 700), (300, 500, 120, 640)])
def test_resize_bicubic_antialiased(H, Hh, Ww, OH, OW):
    """MoGe-1 input resize (v1.py:275): ATen _upsample_bicubic2d_aa (a = -0.5, support 2 max(scale, 1)) - up, down and mixed."""
    img = torch.rand(2, 3, Hh, Ww, generator=torch.Generator().manual_seed(OH))
    ref = F.interpolate(img, (OH, OW), mode="bicubic", align_corners=False, antialias=True)      # CPU reference (ATen)
    assert float((H.resize_bicubic_aa(img, OH, OW).cpu() - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,Hh,Ww,Cc,G", [(2, 24, 30, 256, 1), (2, 24, 30, 256, 8), (1, 111, 77, 128, 4), (3, 50, 41, 32, 1), (1, 97, 130, 64, 2)])
def test_groupnorm_relu(H, prec, B, Hh, Ww, Cc, G):
    """ResidualConvBlock norms (v1.py:44,47): GroupNorm(1, C) and GroupNorm(C / 32, C), eps 1e-5, followed by ReLU; slabs of 2048 pixels per
    block (multi-slab and ragged last slab covered)."""
    g = torch.Generator().manual_seed(Cc + G)
    x = torch.randn(B, Cc, Hh, Ww, generator=g) * 2 + 0.7
    w, b = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    xin = x.half().float() if prec else x
    ref = F.relu(F.group_norm(xin.cuda(), G, w.cuda(), b.cuda(), 1e-5)).permute(0, 2, 3, 1)
    out = H.groupnorm_relu(prec, x.permute(0, 2, 3, 1), w, b, G)
    assert relmax(out, ref) < (2e-5 if prec == 0 else 2e-3)


_ACTS = [F.relu, lambda t: F.leaky_relu(t, 0.2), F.silu, F.elu]


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,Hh,Ww,Cc,norm,act,in_place", [
    (2, 24, 30, 64, "instance", 2, False), (1, 111, 77, 128, "instance", 3, True), (3, 50, 41, 32, "instance", 0, False), (1, 70, 66, 1024, "instance", 1, True),
    (2, 24, 30, 256, "none", 1, False), (1, 97, 130, 40, "none", 2, True), (2, 33, 31, 64, "none", 3, False),
    (2, 24, 30, 256, "group", 2, True), (1, 111, 77, 128, "layer", 3, False), (1, 40, 52, 1024, "group", 1, False), (1, 97, 130, 64, "layer", 0, True)])
def test_residual_block_norm_and_activation(H, prec, B, Hh, Ww, Cc, norm, act, in_place):
    """ResidualConvBlock's [norm ->] activation pairs (modules.py:31-58): InstanceNorm2d (per channel, no affine, eps 1e-5), GroupNorm(1, C),
    GroupNorm(C / 32, C) or no norm, followed by ReLU / LeakyReLU(0.2) / SiLU / ELU; out of place and on its own input buffer (the hidden norm
    of a block runs in place); widths up to 1024 (a 4 x 256 hidden map), multi-slab and ragged last slabs."""
    g = torch.Generator().manual_seed(Cc + act)
    x = torch.randn(B, Cc, Hh, Ww, generator=g) * 2 + 0.7
    xin = (x.half().float() if prec else x).cuda()
    w = b = None
    if norm == "instance":
        y, groups = F.instance_norm(xin, eps=1e-5), Cc
    elif norm == "none":
        y, groups = xin, 0
    else:
        groups = 1 if norm == "layer" else Cc // 32
        w, b = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
        y = F.group_norm(xin, groups, w.cuda(), b.cuda(), 1e-5)
    ref = _ACTS[act](y).permute(0, 2, 3, 1)
    out = H.norm_act(prec, x.permute(0, 2, 3, 1), w, b, groups, act, in_place)
    assert relmax(out, ref) < (2e-5 if prec == 0 else 2e-3)


def test_cast_f16_matches_torch_half_bit_for_bit(H):
    """moge_cast_f16 (what MoGeModel.forward of a half model returns through, v2.py:386-387): round-to-nearest-even like `.half()`, including
    ties, overflow to inf, subnormal halves, signed zeros, inf and NaN."""
    from moge_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(100003, generator=g) * 10.0 ** torch.randint(-9, 6, (100003,), generator=g).float(),
                   torch.tensor([0.0, -0.0, 65504.0, 65519.9, 65520.0, -70000.0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11,
                                 float("inf"), float("-inf"), float("nan")])]).cuda()
    out = torch.empty(x.shape, dtype=torch.float16, device="cuda")
    L.check(L.lib.moge_cast_f16(x.data_ptr(), out.data_ptr(), x.numel(), L.stream_ptr(x.device)))
    torch.cuda.synchronize()
    ref = x.half()
    assert torch.equal(out.view(torch.int16)[~torch.isnan(ref)], ref.view(torch.int16)[~torch.isnan(ref)])
    assert torch.isnan(out[torch.isnan(ref)]).all()
