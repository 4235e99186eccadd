"""GPU: every flavour of conv_pp_kernel the decoder runs (residual add, fused 1x1 side input - both the two-halo-buffer form and the
register form of the 64-channel level -, uv term, bilinear x2 + 3x3 pixel-shuffle resampler) and - in `--experiments` builds of the library
only - the fused residual block of tools/experiments/conv_rb.hip (default-off for two rounds because it is slower: out of the product build),
each against F.conv2d / F.interpolate in fp32 on the SAME fp16-rounded operands, at tile-border sizes and at the decoder's own 480 / 240
maps.  Tolerance as tests/test_hip_gemm_pp.close(): max |err| <= 1e-3 max|ref|, mean |err| <= 1e-4 max|ref| (+ the fp16 output rounding)."""
import pytest
import torch
import torch.nn.functional as F

import os

pytestmark = pytest.mark.gpu
EXPERIMENTS = os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "moge_amd", "lib", "obj", ".experiments"))
needs_experiments = pytest.mark.skipif(not EXPERIMENTS, reason="tools/experiments/conv_rb.hip is compiled by `python -m moge_amd.build --experiments` only")


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests import hip_util
    return hip_util


def r16(t):
    return t.half().float()


def close(out, ref, what=""):
    out, ref = out.double().cpu(), ref.double().cpu()
    scale = float(ref.abs().max().clamp_min(1e-6))
    err = (out - ref).abs()
    # the kernel's output is fp16: half an ulp of the largest value is 2^-11 relative
    assert float(err.max()) <= (1e-3 + 2.0 ** -11) * scale, (what, float(err.max()) / scale)
    assert float(err.mean()) <= 2e-4 * scale, (what, float(err.mean()) / scale)


def conv_ref(x_nhwc, w, b, relu_in=False):
    x = x_nhwc.permute(0, 3, 1, 2).cuda()
    x = F.relu(x) if relu_in else x
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), w.cuda(), b.cuda()).permute(0, 2, 3, 1)


def mk(B, Hh, Ww, Cin, Cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = r16(torch.randn(B, Hh, Ww, Cin, generator=g))
    w = r16(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5)
    b = torch.randn(Cout, generator=g)
    return g, x, w, b


SHAPES = [(2, 23, 45, 64, 64), (1, 16, 16, 64, 64), (3, 17, 70, 64, 64), (2, 17, 31, 128, 128), (1, 33, 21, 256, 256), (1, 40, 48, 128, 128)]


@pytest.mark.parametrize("B,Hh,Ww,Cin,Cout", SHAPES)
def test_conv_residual_add(H, B, Hh, Ww, Cin, Cout):
    """x + conv(h): the second conv of a residual block (modules.py:66); the add is fp16 + fp16 as in the reference's .half() model."""
    g, x, w, b = mk(B, Hh, Ww, Cin, Cout, 1)
    add = r16(torch.randn(B, Hh, Ww, Cout, generator=g))
    ref = r16(conv_ref(x, w, b)) + add.cuda()
    close(H.conv_ex(x, w, b, add=add), ref, "add")


@pytest.mark.parametrize("side_reg", [1, 0])
@pytest.mark.parametrize("B,Hh,Ww,Cin,Cout", SHAPES)
def test_conv_fused_side_input(H, side_reg, B, Hh, Ww, Cin, Cout):
    """conv3x3(x) + W2 . side + bias: the head's `x + in_l(neck_l)` (modules.py:245) fused into the resampler conv.  side_reg = 1: the
    64-channel level's register form (one halo buffer, two workgroups per CU), 0: the two-halo-buffer form for every width."""
    from moge_amd import _lib as L
    g, x, w, b = mk(B, Hh, Ww, Cin, Cout, 2)
    side = r16(torch.randn(B, Hh, Ww, Cin, generator=g))
    sw = r16(torch.randn(Cout, Cin, generator=g) / Cin ** 0.5)
    ref = conv_ref(x, w, b) + torch.einsum("bhwc,oc->bhwo", side.cuda(), sw.cuda())
    L.tune("CONV_SIDE_REG", side_reg)
    try:
        out = H.conv_ex(x, w, b, side=side, side_w=sw)
    finally:
        L.tune("CONV_SIDE_REG", 1)
    close(out, ref, "side")


@pytest.mark.parametrize("relu_in,act", [(False, 0), (True, 1)])
@pytest.mark.parametrize("B,Hh,Ww,Cin,Cout", SHAPES[:4])
def test_conv_uv_term(H, relu_in, act, B, Hh, Ww, Cin, Cout):
    """+ wu u(x) + wv v(y): the (u, v) input channels of the neck's input blocks as a rank-2 epilogue term (v2.py:154-160)."""
    if relu_in and Cin != 64:
        pytest.skip("ReLU prologue + uv: not a decoder combination")
    g, x, w, b = mk(B, Hh, Ww, Cin, Cout, 3)
    wu, wv = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    u = torch.linspace(-0.7, 0.7, Ww)
    v = torch.linspace(-0.6, 0.6, Hh)
    ref = conv_ref(x, w, b, relu_in) + (wu[None, None, None, :] * u[None, None, :, None] + wv[None, None, None, :] * v[None, :, None, None]).cuda()
    if act == 1:
        ref = F.relu(ref)
    if relu_in:            # conv_pp takes ReLU prologue + plain store only; with uv the shape falls to gemm.hip - still the documented semantics
        out = H.conv_ex(x, w, b, relu_in=True, act=act, uv=(wu, wv, -0.7, 0.7, -0.6, 0.6))
    else:
        out = H.conv_ex(x, w, b, act=act, uv=(wu, wv, -0.7, 0.7, -0.6, 0.6))
    close(out, ref, "uv")


@pytest.mark.parametrize("uv", [False, True])
@pytest.mark.parametrize("B,Hh,Ww,Cin,Cout", [(2, 19, 33, 64, 32), (1, 16, 16, 64, 32), (2, 9, 21, 128, 64), (1, 30, 30, 64, 32)])
def test_conv_up2_pixel_shuffle(H, uv, B, Hh, Ww, Cin, Cout):
    """bilinear x2 + 3x3 (modules.py:155-159) as the 4-phase conv on the low-res map + pixel shuffle; uv term at the HIGH resolution."""
    g, x, w, b = mk(B, Hh, Ww, Cin, Cout, 4)
    up = F.interpolate(x.permute(0, 3, 1, 2).cuda(), scale_factor=2, mode="bilinear", align_corners=False)
    ref = F.conv2d(F.pad(up, (1, 1, 1, 1), mode="replicate"), w.cuda(), b.cuda()).permute(0, 2, 3, 1)
    kw = {}
    if uv:
        wu, wv = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
        u = torch.linspace(-0.7, 0.7, 2 * Ww)
        v = torch.linspace(-0.6, 0.6, 2 * Hh)
        ref = ref + (wu[None, None, None, :] * u[None, None, :, None] + wv[None, None, None, :] * v[None, :, None, None]).cuda()
        kw["uv"] = (wu, wv, -0.7, 0.7, -0.6, 0.6)
    out = H.conv_ex(x, w, b, up2=True, **kw)
    # the 4-phase weights are combined in fp32 and rounded ONCE to fp16 (pack_phase_conv_kernel): an extra 2^-11 relative per weight
    out, ref = out.double().cpu(), ref.double().cpu()
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) <= 2.5e-3 * scale
    assert float((out - ref).abs().mean()) <= 3e-4 * scale


@pytest.mark.parametrize("uv", [False, True])
@pytest.mark.parametrize("rows", [1, 3, 4])
@pytest.mark.parametrize("B,Hh,Ww", [(2, 19, 33), (1, 16, 16), (1, 37, 50)])
def test_conv_up2_with_fused_output_conv(H, uv, rows, B, Hh, Ww):
    """The level-4 resampler with the head's 1x1 output conv (modules.py:231) applied inside it: (B, 2H, 2W, 4) fp32 = Wout . fp16(up2 result), the
    32-channel map itself never stored.  Against the unfused kernel's own fp16 output contracted in fp32 with the fp16-rounded weights (exact up
    to fp32 summation order) and against F.conv2d end to end."""
    g, x, w, b = mk(B, Hh, Ww, 64, 32, 8)
    wout = torch.randn(rows, 32, generator=g) / 32 ** 0.5
    kw = {}
    if uv:
        kw["uv"] = (torch.randn(32, generator=g), torch.randn(32, generator=g), -0.7, 0.7, -0.6, 0.6)
    x4 = H.conv_ex(x, w, b, up2=True, **kw)                                   # the unfused path's fp16 map
    got = H.conv_ex(x, w, b, up2=True, dot_w=wout, **kw)
    ref = torch.einsum("bhwc,oc->bhwo", x4.double(), r16(wout).double().cuda())
    scale = float(ref.abs().max())
    assert float((got[..., :rows].double() - ref).abs().max()) <= 2e-6 * scale + 1e-6
    assert float(got[..., rows:].abs().max()) == 0.0 if rows < 4 else True


def resblock_ref(x, w1, b1, w2, b2):
    h = r16(F.relu(conv_ref(x, w1, b1, True)))                       # the intermediate map is fp16 in both the fused and the two-launch path
    xc = h.permute(0, 3, 1, 2)
    y = F.conv2d(F.pad(xc, (1, 1, 1, 1), mode="replicate"), w2.cuda(), b2.cuda()).permute(0, 2, 3, 1)
    return r16(y) + x.cuda()


@needs_experiments
@pytest.mark.parametrize("B,Hh,Ww", [(2, 23, 45), (1, 16, 16), (3, 17, 70), (1, 5, 3), (2, 32, 48), (1, 50, 37), (1, 1, 1)])
def test_fused_residual_block(H, B, Hh, Ww):
    """conv_rb.hip: x + conv2(relu(conv1(relu(x)) + b1)) + b2 in ONE launch, the intermediate tile in LDS.  Against fp32 F.conv2d with the
    intermediate rounded to fp16 (what both paths store), on every border configuration (partial tiles right / bottom, images smaller than
    a tile, 1 x 1)."""
    g, x, w1, b1 = mk(B, Hh, Ww, 64, 64, 5)
    w2 = r16(torch.randn(64, 64, 3, 3, generator=g) / (9 * 64) ** 0.5)
    b2 = torch.randn(64, generator=g)
    ref = resblock_ref(x, w1, b1, w2, b2)
    out = H.conv_ex(x, w1, b1, w2=w2, bias2=b2)
    close(out, ref, "resblock")


@needs_experiments
@pytest.mark.parametrize("grid", [3, 13])
def test_fused_residual_block_equals_two_launch_path_bitwise(H, grid):
    """The fused kernel accumulates each conv in the two-launch kernels' order (same MFMA form, taps 0..8, two K-steps of 32) and rounds the
    intermediate and the output exactly where they do: BIT-identical results, with small workgroup caps (several tiles per workgroup, the
    fewer-workgroups-than-XCDs split) and with one tile per workgroup."""
    from moge_amd import _lib as L
    B, Hh, Ww = 3, 50, 37
    g, x, w1, b1 = mk(B, Hh, Ww, 64, 64, 6)
    w2 = r16(torch.randn(64, 64, 3, 3, generator=g) / (9 * 64) ** 0.5)
    b2 = torch.randn(64, generator=g)
    h = H.conv_ex(x, w1, b1, relu_in=True, act=1)
    two = H.conv_ex(h, w2, b2, add=x)
    outs = []
    for cap in (grid, 1 << 20):
        L.tune("CONV_GRID", cap)
        try:
            outs.append(H.conv_ex(x, w1, b1, w2=w2, bias2=b2))
        finally:
            L.tune("CONV_GRID", 0)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], two)


@pytest.mark.parametrize("what", ["add", "side", "resblock", "up2"])
def test_decoder_sized_maps(H, what):
    """The level-3 map of the bench workload (480 x 480 x 64, two images): full persistent walks, XCD ranges, tile borders at 480 = 30 x 16."""
    B, Hh, Ww = 2, 480, 480
    g, x, w, b = mk(B, Hh, Ww, 64, 64 if what != "up2" else 32, 7)
    if what == "add":
        add = r16(torch.randn(B, Hh, Ww, 64, generator=g))
        close(H.conv_ex(x, w, b, add=add), r16(conv_ref(x, w, b)) + add.cuda(), what)
    elif what == "side":
        side = r16(torch.randn(B, Hh, Ww, 64, generator=g))
        sw = r16(torch.randn(64, 64, generator=g) / 8)
        close(H.conv_ex(x, w, b, side=side, side_w=sw), conv_ref(x, w, b) + torch.einsum("bhwc,oc->bhwo", side.cuda(), sw.cuda()), what)
    elif what == "resblock":
        if not EXPERIMENTS:
            pytest.skip("fused residual block: --experiments builds only")
        w2 = r16(torch.randn(64, 64, 3, 3, generator=g) / 24)
        b2 = torch.randn(64, generator=g)
        close(H.conv_ex(x, w, b, w2=w2, bias2=b2), resblock_ref(x, w, b, w2, b2), what)
    else:
        up = F.interpolate(x.permute(0, 3, 1, 2).cuda(), scale_factor=2, mode="bilinear", align_corners=False)
        ref = F.conv2d(F.pad(up, (1, 1, 1, 1), mode="replicate"), w.cuda(), b.cuda()).permute(0, 2, 3, 1)
        out = H.conv_ex(x, w, b, up2=True).double().cpu()
        ref = ref.double().cpu()
        assert float((out - ref).abs().max()) <= 2.5e-3 * float(ref.abs().max())


# ---- fused ConvTranspose2d(k2, s2) + 3x3 (conv_pp.hip CT3 + ct3_border_kernel; round 6) ---------------------------------------------------------------
def ct3_ref(x_nhwc, wt, bt, w3, b3, side=None, side_w=None, uv=None):
    """modules.py:160-165 in fp32 on the same operands: ConvTranspose2d -> replicate-padded 3x3 [+ 1x1 side input, + uv term]."""
    x = x_nhwc.permute(0, 3, 1, 2).cuda()
    hi = F.conv_transpose2d(x, wt.cuda(), bt.cuda(), stride=2)
    y = F.conv2d(F.pad(hi, (1, 1, 1, 1), mode="replicate"), w3.cuda(), b3.cuda())
    if side is not None:
        y = y + F.conv2d(side.permute(0, 3, 1, 2).cuda(), side_w.cuda()[:, :, None, None])
    if uv is not None:
        wu, wv, u0, u1, v0, v1 = uv
        Ho, Wo = y.shape[-2:]
        u = torch.linspace(u0, u1, Wo, device="cuda")
        v = torch.linspace(v0, v1, Ho, device="cuda")
        y = y + wu.cuda()[None, :, None, None] * u[None, None, None, :] + wv.cuda()[None, :, None, None] * v[None, None, :, None]
    return y.permute(0, 2, 3, 1)


def mk_ct3(B, Hh, Ww, Cout, seed):
    Cin = 2 * Cout
    g = torch.Generator().manual_seed(seed)
    x = r16(torch.randn(B, Hh, Ww, Cin, generator=g))
    wt = torch.randn(Cin, Cout, 2, 2, generator=g) / Cin ** 0.5
    bt = torch.randn(Cout, generator=g) * 0.5
    w3 = torch.randn(Cout, Cout, 3, 3, generator=g) / (9 * Cout) ** 0.5
    b3 = torch.randn(Cout, generator=g)
    return g, x, wt, bt, w3, b3


CT3_SHAPES = [(2, 17, 31, 128), (1, 16, 16, 128), (1, 33, 21, 64), (2, 5, 70, 64), (1, 1, 1, 128), (1, 2, 19, 64), (1, 40, 48, 128)]


@pytest.mark.parametrize("B,Hh,Ww,Cout", CT3_SHAPES)
def test_fused_convt_conv3_matches_the_pair(H, B, Hh, Ww, Cout):
    """The composed 4-phase conv + its border ring against ConvTranspose2d -> replicate 3x3 in fp32: interior AND every border / corner pixel (tile-border
    sizes, one-pixel and two-row maps).  Without the border pass the outermost ring must be the ONLY place that differs (the composed conv is a reflecting pad)."""
    g, x, wt, bt, w3, b3 = mk_ct3(B, Hh, Ww, Cout, 31)
    ref = ct3_ref(x, wt, bt, w3, b3)
    out = H.ct3(x, wt, bt, w3, b3)
    close(out, ref, "ct3")
    raw = H.ct3(x, wt, bt, w3, b3, no_border=True)
    inner_o, inner_r = raw[:, 1:-1, 1:-1], ref[:, 1:-1, 1:-1]
    if inner_r.numel():
        scale = float(ref.abs().max())
        assert float((inner_o.cpu().double() - inner_r.cpu().double()).abs().max()) <= (1e-3 + 2.0 ** -11) * scale, "interior differs without the border pass"
    ring = torch.ones(ref.shape[1:3], dtype=torch.bool)
    ring[1:-1, 1:-1] = False
    d = (raw.cpu() - ref.cpu()).abs().amax(dim=(0, 3))
    if min(Hh, Ww) > 1:
        assert float(d[ring].max()) > 10 * float(d[~ring].max() if (~ring).any() else 0.0), "the un-corrected ring should show the reflect / replicate difference"


@pytest.mark.parametrize("B,Hh,Ww,Cout", [(2, 17, 31, 128), (1, 33, 21, 64), (1, 24, 24, 64)])
def test_fused_convt_conv3_with_side_input_and_uv(H, B, Hh, Ww, Cout):
    """the heads' form (fused 1x1 input block on the HIGH-res neck map, modules.py:245; Cout = 128 only: the 64-channel form runs two phases per workgroup and
    refuses a side image - the decoder then keeps the unfused pair) and the neck's (uv term at the output resolution)"""
    g, x, wt, bt, w3, b3 = mk_ct3(B, Hh, Ww, Cout, 32)
    side = r16(torch.randn(B, 2 * Hh, 2 * Ww, Cout, generator=g))
    side_w = r16(torch.randn(Cout, Cout, generator=g) / Cout ** 0.5)
    if Cout == 128:
        out = H.ct3(x, wt, bt, w3, b3, side=side, side_w=side_w)
        close(out, ct3_ref(x, wt, bt, w3, b3, side=side, side_w=side_w), "ct3 + side")
    else:
        from moge_amd._lib import MogeError
        with pytest.raises(MogeError):
            H.ct3(x, wt, bt, w3, b3, side=side, side_w=side_w)
    uv = (torch.randn(Cout, generator=g), torch.randn(Cout, generator=g), -0.7, 0.7, -0.6, 0.6)
    out = H.ct3(x, wt, bt, w3, b3, uv=uv)
    close(out, ct3_ref(x, wt, bt, w3, b3, uv=uv), "ct3 + uv")


def test_fused_convt_conv3_at_the_decoder_sizes(H):
    """the two resamplers of the released layout that take the fused form: 256 -> 128 at 120 x 120 and 128 -> 64 at 240 x 240 (one image)"""
    for Hh, Cout in ((120, 128), (240, 64)):
        g, x, wt, bt, w3, b3 = mk_ct3(1, Hh, Hh, Cout, 33)
        close(H.ct3(x, wt, bt, w3, b3), ct3_ref(x, wt, bt, w3, b3), f"ct3 {Hh}")
