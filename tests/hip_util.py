"""ctypes helpers for the per-kernel C-ABI test entry points (tests only)."""
import torch

from moge_amd import _lib as L


def _p(t):
    return None if t is None else t.data_ptr()


def _f(t):
    return t.detach().to("cuda", torch.float32).contiguous()


def st():
    return L.stream_ptr()


def gemm(prec, A, W, bias=None, act=0):
    A, W = _f(A), _f(W)
    bias = None if bias is None else _f(bias)
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty((M, N), device="cuda", dtype=torch.float32)
    L.check(L.lib.moge_test_gemm(prec, _p(A), _p(W), _p(bias), _p(C), M, N, K, act, st()))
    return C


TG_STORE, TG_RESID, TG_QKV, TG_CONVT = 0, 1, 2, 3


def gemm_ex(kind, A, W, bias, prec=1, act=0, ln_mr=None, ln_c=None, uv=None, pix=None, Cout=0, xres=None, gamma=None, want_x16=False,
            nh=0, Ntok=0, qscale=1.0, x16_stream=None, want_part=True):
    """One GEMM through a fused epilogue (moge_test_gemm_ex).  All tensors fp32 on the GPU.  Returns a dict of outputs."""
    import ctypes as C
    A, W = _f(A), _f(W)
    M, K = A.shape
    N = W.shape[0]
    a = L.TestGemmArgs()
    keep = [A, W]

    def dev(t):
        if t is None:
            return None
        t = _f(t)
        keep.append(t)
        return t.data_ptr()

    a.precision, a.kind, a.act, a.M, a.N, a.K = prec, kind, act, M, N, K
    a.A, a.W, a.bias = A.data_ptr(), W.data_ptr(), dev(bias)
    a.ln_mr, a.ln_c = dev(ln_mr), dev(ln_c)
    if pix is not None:
        a.pixW, a.pixH = pix
    if uv is not None:
        wu, wv, u0, u1, v0, v1 = uv
        a.wu, a.wv, a.u0, a.u1, a.v0, a.v1 = dev(wu), dev(wv), u0, u1, v0, v1
    out = {}
    if kind in (TG_STORE, TG_CONVT):
        out["out"] = torch.empty((M, N), device="cuda", dtype=torch.float32)
        a.out, a.Cout = out["out"].data_ptr(), Cout
    elif kind == TG_RESID and x16_stream is not None:
        # fp16 residual stream of a `.half()` model (EPK_RESID16): x16 in / out (fp32 values that are exact fp16 numbers), xres = NULL
        out["x16"] = _f(x16_stream).clone()
        a.x16_out, a.gamma = out["x16"].data_ptr(), dev(gamma)
        if want_part:
            out["ln_part"] = torch.empty((M, N // 32, 2), device="cuda", dtype=torch.float32)
            a.ln_part_out = out["ln_part"].data_ptr()
    elif kind == TG_RESID:
        out["xres"] = _f(xres).clone()
        a.xres, a.gamma = out["xres"].data_ptr(), dev(gamma)
        if want_x16:
            out["x16"] = torch.empty((M, N), device="cuda", dtype=torch.float32)
            out["ln_part"] = torch.empty((M, N // 32, 2), device="cuda", dtype=torch.float32)
            a.x16_out, a.ln_part_out = out["x16"].data_ptr(), out["ln_part"].data_ptr()
    elif kind == TG_QKV:
        B = M // Ntok
        for k in ("q", "k", "v"):
            out[k] = torch.empty((B, nh, Ntok, 64), device="cuda", dtype=torch.float32)
        a.q_out, a.k_out, a.v_out = out["q"].data_ptr(), out["k"].data_ptr(), out["v"].data_ptr()
        a.nh, a.Ntok, a.qscale = nh, Ntok, qscale
    L.check(L.lib.moge_test_gemm_ex(C.byref(a), st()))
    return out


def layernorm(prec, x, w, b):
    x, w, b = _f(x), _f(w), _f(b)
    y = torch.empty_like(x)
    L.check(L.lib.moge_test_layernorm(prec, _p(x), _p(w), _p(b), _p(y), x.shape[0], x.shape[1], st()))
    return y


def attention(prec, q, k, v):
    q, k, v = _f(q), _f(k), _f(v)
    B, nh, N, _ = q.shape
    o = torch.empty((B, N, nh * 64), device="cuda", dtype=torch.float32)
    L.check(L.lib.moge_test_attention(prec, _p(q), _p(k), _p(v), _p(o), B, nh, N, st()))
    return o


def conv3x3(prec, x_nhwc, w, bias, relu_in=False, up2=False):
    x, w, bias = _f(x_nhwc), _f(w), _f(bias)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
    y = torch.empty((B, Ho, Wo, Cout), device="cuda", dtype=torch.float32)
    L.check(L.lib.moge_test_conv3x3(prec, _p(x), _p(w), _p(bias), _p(y), B, H, W, Cin, Cout, (1 if relu_in else 0) | (2 if up2 else 0), st()))
    return y


def convt2x2(prec, x_nhwc, w, bias):
    x, w, bias = _f(x_nhwc), _f(w), _f(bias)
    B, H, W, Cin = x.shape
    Cout = w.shape[1]
    y = torch.empty((B, 2 * H, 2 * W, Cout), device="cuda", dtype=torch.float32)
    L.check(L.lib.moge_test_convt2x2(prec, _p(x), _p(w), _p(bias), _p(y), B, H, W, Cin, Cout, st()))
    return y


def preprocess(img, rows, cols):
    img = _f(img)
    B, _, H, W = img.shape
    y = torch.empty((B, 3, rows * 14, cols * 14), device="cuda", dtype=torch.float32)
    L.check(L.lib.moge_test_preprocess(_p(img), _p(y), B, H, W, rows, cols, st()))
    return y


def posembed(pos, rows, cols):
    pos = _f(pos)
    D = pos.shape[-1]
    y = torch.empty((1 + rows * cols, D), device="cuda", dtype=torch.float32)
    L.check(L.lib.moge_test_posembed(_p(pos), _p(y), D, rows, cols, st()))
    return y


def recover(points, mask, focal=None):
    points = _f(points)
    B, H, W, _ = points.shape
    m = None if mask is None else mask.to("cuda", torch.uint8).contiguous()
    f_in = None if focal is None else _f(focal)
    f = torch.empty((B,), device="cuda", dtype=torch.float32)
    s = torch.empty((B,), device="cuda", dtype=torch.float32)
    status = torch.zeros((1,), device="cuda", dtype=torch.int32)
    L.check(L.lib.moge_test_recover(_p(points), _p(m), _p(f_in), B, H, W, _p(f), _p(s), _p(status), st()))
    return f, s, int(status.item())


def resize_bicubic_aa(img, OH, OW):
    img = _f(img)
    B, _, H, W = img.shape
    y = torch.empty((B, 3, OH, OW), device="cuda", dtype=torch.float32)
    L.check(L.lib.moge_test_resize_bicubic_aa(_p(img), _p(y), B, H, W, OH, OW, st()))
    return y


def groupnorm_relu(prec, x_nhwc, gamma, beta, groups):
    x, gamma, beta = _f(x_nhwc), _f(gamma), _f(beta)
    B, H, W, Cc = x.shape
    y = torch.empty_like(x)
    L.check(L.lib.moge_test_groupnorm_relu(prec, _p(x), _p(gamma), _p(beta), _p(y), B, H, W, Cc, groups, st()))
    return y


def norm_act(prec, x_nhwc, gamma, beta, groups, act, in_place=False):
    """act(norm(x)) of a v2 residual block: groups 0 = no norm, C = InstanceNorm2d (gamma = beta = None)."""
    x = _f(x_nhwc)
    B, H, W, Cc = x.shape
    y = torch.empty_like(x)
    g = _f(gamma) if gamma is not None else None
    b = _f(beta) if beta is not None else None
    L.check(L.lib.moge_test_norm_act(prec, _p(x), _p(g) if g is not None else None, _p(b) if b is not None else None, _p(y), B, H, W, Cc, groups, act, int(in_place), st()))
    return y


def conv_ex(x_nhwc, w, bias, prec=1, relu_in=False, act=0, add=None, side=None, side_w=None, uv=None, up2=False, w2=None, bias2=None, dot_w=None):
    """One 3x3 conv through the pieces the decoder fuses into it (moge_test_conv_ex).  uv = (wu, wv, u0, u1, v0, v1) at the OUTPUT resolution;
    w2 / bias2 select the fused residual block (conv_rb.hip)."""
    import ctypes as C
    keep = []

    def dev(t):
        if t is None:
            return None
        t = _f(t)
        keep.append(t)
        return t.data_ptr()

    x = _f(x_nhwc)
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
    y = torch.empty((B, Ho, Wo, 4 if dot_w is not None else Cout), device="cuda", dtype=torch.float32)
    a = L.TestConvArgs()
    a.precision, a.B, a.H, a.W, a.Cin, a.Cout = prec, B, H, W, Cin, Cout
    a.relu_in, a.act, a.up2 = int(bool(relu_in)), act, int(bool(up2))
    a.x, a.w, a.bias = x.data_ptr(), dev(w), dev(bias)
    a.add, a.side, a.side_w = dev(add), dev(side), dev(side_w)
    if uv is not None:
        wu, wv, u0, u1, v0, v1 = uv
        a.wu, a.wv, a.u0, a.u1, a.v0, a.v1 = dev(wu), dev(wv), u0, u1, v0, v1
    a.w2, a.bias2 = dev(w2), dev(bias2)
    if dot_w is not None:
        a.dot_w, a.dot_rows = dev(dot_w), int(dot_w.shape[0])
    a.y = y.data_ptr()
    L.check(L.lib.moge_test_conv_ex(C.byref(a), st()))
    return y


def ct3(x_nhwc, wt, bt, w3, b3, side=None, side_w=None, uv=None, no_border=False):
    """ConvTranspose2d(k2, s2) + 3x3 replicate conv through the fused fp16 path (moge_test_ct3): x (B,H,W,Cin), wt (Cin,Cout,2,2), w3 (Cout,Cout,3,3)
    -> (B,2H,2W,Cout).  side (B,2H,2W,Cout) / side_w (Cout,Cout): the fused 1x1 input block; uv = (wu, wv, u0, u1, v0, v1) at the output resolution."""
    import ctypes as C
    keep = []

    def dev(t):
        if t is None:
            return None
        t = _f(t)
        keep.append(t)
        return t.data_ptr()

    x = _f(x_nhwc)
    B, H, W, Cin = x.shape
    Cout = w3.shape[0]
    y = torch.empty((B, 2 * H, 2 * W, Cout), device="cuda", dtype=torch.float32)
    a = L.TestCt3Args()
    a.precision, a.B, a.H, a.W, a.Cin, a.Cout, a.no_border = 1, B, H, W, Cin, Cout, int(bool(no_border))
    a.x, a.wt, a.bt, a.w3, a.b3 = x.data_ptr(), dev(wt), dev(bt), dev(w3), dev(b3)
    a.side, a.side_w = dev(side), dev(side_w)
    if uv is not None:
        wu, wv, u0, u1, v0, v1 = uv
        a.wu, a.wv, a.u0, a.u1, a.v0, a.v1 = dev(wu), dev(wv), u0, u1, v0, v1
    a.y = y.data_ptr()
    L.check(L.lib.moge_test_ct3(C.byref(a), st()))
    return y
