// Per-kernel test entry points of the C ABI (include/moge_hip.h, "moge_test_*").  tests/ only: they allocate
// temporaries with hipMalloc, convert fp32 <-> storage type around the kernel under test and synchronise.
#include "launchers.h"
#include "../../include/moge_hip.h"
#include <vector>
#include <type_traits>
#include <cstring>
#include <cstdio>

#define TCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "moge_test: %s -> %s\n", #x, hipGetErrorString(e_)); return MOGE_ERR_HIP; } } while (0)
#define TL(x) do { int e_ = (x); if (e_ != 0) { fprintf(stderr, "moge_test: %s -> launch error %d\n", #x, e_); return e_ < 0 ? MOGE_ERR_INVALID : MOGE_ERR_HIP; } } while (0)

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

template <typename T> static int to_t(const float* src, void* dst, long n, hipStream_t st) { return launch_convert<float, T>(src, dst, n, st); }
template <typename T> static int from_t(const void* src, float* dst, long n, hipStream_t st);
template <> int from_t<f16>(const void* src, float* dst, long n, hipStream_t st) { return launch_convert<f16, float>(src, dst, n, st); }
template <> int from_t<float>(const void* src, float* dst, long n, hipStream_t st) { return launch_convert<float, float>(src, dst, n, st); }

template <typename T>
static int t_gemm(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int act, hipStream_t st) {
    const int CH = TT<T>::CH;
    const int Kp = (K + CH - 1) / CH * CH;
    DevBuf a, w, c;
    TCHK(a.alloc((size_t)M * Kp * sizeof(T))); TCHK(w.alloc((size_t)N * Kp * sizeof(T))); TCHK(c.alloc((size_t)M * N * sizeof(T)));
    TCHK(hipMemsetAsync(a.p, 0, (size_t)M * Kp * sizeof(T), st)); TCHK(hipMemsetAsync(w.p, 0, (size_t)N * Kp * sizeof(T), st));
    TL(launch_repack<T>(A, a.p, M, 1, 1, K, K, 0, 0, 1, Kp, 0, 0, st));
    TL(launch_repack<T>(W, w.p, N, 1, 1, K, K, 0, 0, 1, Kp, 0, 0, st));
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.a = a.p; g.lda = Kp; g.w = w.p; g.ldw = Kp; g.M = M; g.N = N; g.K = Kp;
    g.epi = EPI_STORE; g.act = act; g.bias = bias; g.out = c.p; g.ldc = N;
    TL(launch_gemm<T>(g, AMODE_LINEAR, st));
    TL(from_t<T>(c.p, C, (long)M * N, st));
    TCHK(hipStreamSynchronize(st));
    return 0;
}

template <typename T>
static int t_gemm_ex(const moge_test_gemm_args& a, hipStream_t st) {
    const int M = a.M, N = a.N, K = a.K;
    if (K % TT<T>::CH) return MOGE_ERR_INVALID;
    DevBuf ab, wb, ob, x16, qb, kb, vb;
    TCHK(ab.alloc((size_t)M * K * sizeof(T))); TCHK(wb.alloc((size_t)N * K * sizeof(T)));
    TL(to_t<T>(a.A, ab.p, (long)M * K, st));
    TL(to_t<T>(a.W, wb.p, (long)N * K, st));
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.a = ab.p; g.lda = K; g.w = wb.p; g.ldw = K; g.M = M; g.N = N; g.K = K;
    g.bias = a.bias; g.act = a.act; g.ln_mr = a.ln_mr; g.ln_c = a.ln_c;
    g.pixW = a.pixW; g.pixH = a.pixH;
    const size_t mn = (size_t)M * N;
    switch (a.kind) {
    case MOGE_TG_STORE:
        TCHK(ob.alloc(mn * sizeof(T)));
        g.epi = EPI_STORE; g.out = ob.p; g.ldc = N;
        if (a.wu) {
            g.uv.wu = a.wu; g.uv.wv = a.wv; g.uv.u0 = a.u0; g.uv.u1 = a.u1; g.uv.v0 = a.v0; g.uv.v1 = a.v1;
            g.uv.ustep = a.pixW > 1 ? (a.u1 - a.u0) / (float)(a.pixW - 1) : 0.f;
            g.uv.vstep = a.pixH > 1 ? (a.v1 - a.v0) / (float)(a.pixH - 1) : 0.f;
        }
        break;
    case MOGE_TG_RESID:
        g.epi = EPI_RESID; g.xres = a.xres; g.ldc = N; g.gamma = a.gamma;
        if (!a.xres) {
            // fp16 residual stream (`.half()` models): x16_out is IN / OUT - its fp32 values are rounded to fp16, updated in place by the
            // epilogue and returned as fp32; ln_part_out (optional) receives the statistics of the ROUNDED result
            if (!std::is_same<T, f16>::value || !a.x16_out) return MOGE_ERR_INVALID;
            TCHK(x16.alloc(mn * sizeof(f16)));
            TL((launch_convert<float, f16>(a.x16_out, x16.p, (long)mn, st)));
            g.x16 = x16.p; g.ln_part = a.ln_part_out;
        } else if (a.x16_out) {
            if (!std::is_same<T, f16>::value || !a.ln_part_out) return MOGE_ERR_INVALID;
            TCHK(x16.alloc(mn * sizeof(f16)));
            g.x16 = x16.p; g.ln_part = a.ln_part_out;
        }
        break;
    case MOGE_TG_QKV: {
        const int D = a.nh * 64;
        if (N != 3 * D || M % a.Ntok) return MOGE_ERR_INVALID;
        const size_t n = (size_t)M * D;
        TCHK(qb.alloc(n * sizeof(T))); TCHK(kb.alloc(n * sizeof(T))); TCHK(vb.alloc(n * sizeof(T)));
        g.epi = EPI_QKV; g.q = qb.p; g.k = kb.p; g.vT = vb.p; g.v_rowmajor = 1;
        g.nh = a.nh; g.D = D; g.Ntok = a.Ntok; g.Npad = (a.Ntok + 63) / 64 * 64; g.qscale = a.qscale;
        break;
    }
    case MOGE_TG_CONVT:
        TCHK(ob.alloc(mn * sizeof(T)));
        g.epi = EPI_CONVT; g.out = ob.p; g.Cout = a.Cout;
        break;
    default: return MOGE_ERR_INVALID;
    }
    TL(launch_gemm<T>(g, AMODE_LINEAR, st));
    if (a.kind == MOGE_TG_STORE || a.kind == MOGE_TG_CONVT) TL(from_t<T>(ob.p, a.out, (long)mn, st));
    if (a.kind == MOGE_TG_RESID && a.x16_out) TL((launch_convert<f16, float>(x16.p, a.x16_out, (long)mn, st)));
    if (a.kind == MOGE_TG_QKV) {
        const long n = (long)M * a.nh * 64;
        TL(from_t<T>(qb.p, a.q_out, n, st)); TL(from_t<T>(kb.p, a.k_out, n, st)); TL(from_t<T>(vb.p, a.v_out, n, st));
    }
    TCHK(hipStreamSynchronize(st));
    return 0;
}

template <typename T>
static int t_attention(const float* q, const float* k, const float* v, float* o, int B, int nh, int N, hipStream_t st) {
    const int Npad = (N + 63) / 64 * 64;
    const size_t n = (size_t)B * nh * N * 64;
    DevBuf qb, kb, vb, ob, qs;
    TCHK(qb.alloc(n * sizeof(T))); TCHK(kb.alloc(n * sizeof(T))); TCHK(vb.alloc((size_t)B * nh * 64 * Npad * sizeof(T))); TCHK(ob.alloc(n * sizeof(T)));
    TCHK(qs.alloc(n * sizeof(float)));
    TCHK(hipMemsetAsync(vb.p, 0, (size_t)B * nh * 64 * Npad * sizeof(T), st));
    // q pre-scale by log2(e)/8 (what the QKV epilogue does): reuse repack with a scaled copy through host-free path
    {
        std::vector<float> hq(n);
        TCHK(hipMemcpyAsync(hq.data(), q, n * sizeof(float), hipMemcpyDeviceToHost, st));
        TCHK(hipStreamSynchronize(st));
        const float sc = 0.125f * 1.4426950408889634f;
        for (auto& x : hq) x *= sc;
        TCHK(hipMemcpyAsync(qs.p, hq.data(), n * sizeof(float), hipMemcpyHostToDevice, st));
        TCHK(hipStreamSynchronize(st));
    }
    TL(to_t<T>((const float*)qs.p, qb.p, (long)n, st));
    TL(to_t<T>(k, kb.p, (long)n, st));
    // v (B*nh, N, 64) -> vT (B*nh, 64, Npad)
    TL(launch_repack<T>(v, vb.p, B * nh, 64, 1, N, (long)N * 64, 1, 0, 64, (long)64 * Npad, Npad, 0, st));
    if (std::is_same<T, f16>::value && moge_tune_get("ATTN_PP", 1)) {
        DevBuf vr;                                   // row-major V for the fp16 throughput kernel
        TCHK(vr.alloc(n * sizeof(T)));
        TL(to_t<T>(v, vr.p, (long)n, st));
        DevBuf aws;                                  // stream-K form (sub-round grids; ATTN_SK = 2 forces it): counters zeroed once
        const size_t awb = attention_pp_ws_bytes(B, nh, N);
        if (awb) { TCHK(aws.alloc(awb)); TCHK(hipMemsetAsync(aws.p, 0, attention_pp_ws_counter_bytes(B, nh, N), st)); }
        TL(launch_attention_pp(qb.p, kb.p, vr.p, ob.p, B, nh, N, st, awb ? aws.p : nullptr, awb));
        if (awb && moge_tune_get("ATTN_SK_TWICE", 0)) TL(launch_attention_pp(qb.p, kb.p, vr.p, ob.p, B, nh, N, st, aws.p, awb));      // tests: the counters are left zero
        TL(from_t<T>(ob.p, o, (long)n, st));
        TCHK(hipStreamSynchronize(st));
        return 0;
    }
    TL(launch_attention<T>(qb.p, kb.p, vb.p, ob.p, B, nh, N, Npad, st));
    TL(from_t<T>(ob.p, o, (long)n, st));
    TCHK(hipStreamSynchronize(st));
    return 0;
}

template <typename T>
static int t_conv3(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, int relu_in, int up2, hipStream_t st) {
    // for up2, (H,W) are the INPUT dims and the output is (2H,2W): bilinear x2 + 3x3 as the 4-phase low-res conv
    const int Ho = up2 ? 2 * H : H, Wo = up2 ? 2 * W : W;
    DevBuf xb, wb, yb, bb;
    const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * Ho * Wo * Cout;
    TCHK(xb.alloc(nx * sizeof(T))); TCHK(wb.alloc((size_t)4 * Cout * 9 * Cin * sizeof(T))); TCHK(yb.alloc(ny * sizeof(T))); TCHK(bb.alloc(4 * Cout * sizeof(float)));
    TL(to_t<T>(x, xb.p, (long)nx, st));
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.a = xb.p; g.H = H; g.W = W; g.C = Cin; g.relu_in = relu_in;
    g.w = wb.p; g.ldw = 9 * Cin; g.M = B * H * W; g.K = 9 * Cin; g.out = yb.p; g.pixW = W; g.pixH = H;
    if (up2) {
        TL(launch_pack_phase_conv<T>(w, wb.p, Cout, Cin, st));
        TL(launch_repack<float>(bias, bb.p, 4, 1, 1, Cout, 0, 0, 0, 1, Cout, 0, 0, st));
        g.N = 4 * Cout; g.epi = EPI_CONVT; g.bias = (const float*)bb.p; g.Cout = Cout;
    } else {
        TL(launch_repack<T>(w, wb.p, Cout, 9, 1, Cin, (long)Cin * 9, 1, 0, 9, (long)9 * Cin, Cin, 0, st));
        g.N = Cout; g.epi = EPI_STORE; g.bias = bias; g.ldc = Cout;
    }
    TL(launch_gemm<T>(g, AMODE_CONV3, st));
    TL(from_t<T>(yb.p, y, (long)ny, st));
    TCHK(hipStreamSynchronize(st));
    return 0;
}

// One 3x3 conv through the optional pieces the decoder fuses into it (conv_pp.hip flavours; gemm.hip takes the shapes / precision conv_pp
// does not): ReLU prologue, activation, residual add, fused 1x1 side input, uv term, pixel-shuffle resampler (up2), fused residual block.
template <typename T>
static int t_conv_ex(const moge_test_conv_args& a, hipStream_t st) {
    const int B = a.B, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, up2 = a.up2;
    const int Ho = up2 ? 2 * H : H, Wo = up2 ? 2 * W : W;
    const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * Ho * Wo * Cout;
    DevBuf xb, wb, yb, bb, addb, sideb, sidew, w2b;
    TCHK(xb.alloc(nx * sizeof(T))); TCHK(wb.alloc((size_t)4 * Cout * 9 * Cin * sizeof(T))); TCHK(yb.alloc(ny * sizeof(T))); TCHK(bb.alloc(4 * Cout * sizeof(float)));
    TL(to_t<T>(a.x, xb.p, (long)nx, st));
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.a = xb.p; g.H = H; g.W = W; g.C = Cin; g.relu_in = a.relu_in;
    g.w = wb.p; g.ldw = 9 * Cin; g.M = B * H * W; g.K = 9 * Cin; g.out = yb.p; g.pixW = W; g.pixH = H; g.act = a.act;
    DevBuf dtab, dout;
    if (a.dot_w) {
        if (!up2 || !std::is_same<T, f16>::value || Cin != 64 || Cout != 32 || a.dot_rows < 1 || a.dot_rows > 4) return MOGE_ERR_INVALID;
        TCHK(dtab.alloc(1024)); TCHK(dout.alloc((size_t)B * Ho * Wo * 4 * sizeof(float)));
        TL(launch_pack_dot_table(a.dot_w, a.dot_rows, 1, dtab.p, st));
        g.dot_tab = dtab.p; g.dot_nd = 1; g.dot_out = (float*)dout.p;
    }
    if (up2) {
        if (a.add || a.side || a.w2) return MOGE_ERR_INVALID;
        TL(launch_pack_phase_conv<T>(a.w, wb.p, Cout, Cin, st));
        TL(launch_repack<float>(a.bias, bb.p, 4, 1, 1, Cout, 0, 0, 0, 1, Cout, 0, 0, st));
        g.N = 4 * Cout; g.epi = EPI_CONVT; g.bias = (const float*)bb.p; g.Cout = Cout;
    } else {
        TL(launch_repack<T>(a.w, wb.p, Cout, 9, 1, Cin, (long)Cin * 9, 1, 0, 9, (long)9 * Cin, Cin, 0, st));
        g.N = Cout; g.epi = EPI_STORE; g.bias = a.bias; g.ldc = Cout;
    }
    if (a.wu) {
        g.uv.wu = a.wu; g.uv.wv = a.wv; g.uv.u0 = a.u0; g.uv.u1 = a.u1; g.uv.v0 = a.v0; g.uv.v1 = a.v1;
        g.uv.ustep = Wo > 1 ? (a.u1 - a.u0) / (float)(Wo - 1) : 0.f;
        g.uv.vstep = Ho > 1 ? (a.v1 - a.v0) / (float)(Ho - 1) : 0.f;
    }
    if (a.add) {
        TCHK(addb.alloc(ny * sizeof(T)));
        TL(to_t<T>(a.add, addb.p, (long)ny, st));
        g.add = addb.p; g.ldadd = Cout;
    }
    if (a.side) {
        if (!a.side_w || Cin != Cout) return MOGE_ERR_INVALID;
        TCHK(sideb.alloc(nx * sizeof(T))); TCHK(sidew.alloc((size_t)Cout * Cin * sizeof(T)));
        TL(to_t<T>(a.side, sideb.p, (long)nx, st));
        TL(to_t<T>(a.side_w, sidew.p, (long)Cout * Cin, st));
        g.a2 = sideb.p; g.w2 = sidew.p;
        if (!std::is_same<T, f16>::value || !conv_pp_eligible(g)) return MOGE_ERR_INVALID;      // only the halo kernel fuses the side input
    }
    if (a.w2) {                                   // fused residual block (modules.py:47-68): y = x + conv2(relu(conv1(relu(x)) + b1)) + b2
        if (!std::is_same<T, f16>::value || up2 || a.side || a.add || a.wu || Cin != Cout || !a.bias2) return MOGE_ERR_INVALID;
        TCHK(w2b.alloc((size_t)Cout * 9 * Cin * sizeof(T)));
        TL(launch_repack<T>(a.w2, w2b.p, Cout, 9, 1, Cin, (long)Cin * 9, 1, 0, 9, (long)9 * Cin, Cin, 0, st));
        GemmArgs r = g;
        r.relu_in = 1; r.act = ACT_NONE; r.rb_w2 = w2b.p; r.rb_bias2 = a.bias2; r.add = xb.p; r.ldadd = Cin;
#ifdef MOGE_EXPERIMENTS
        if (!conv_rb_eligible(r)) return MOGE_ERR_INVALID;
        TL(launch_conv_rb(r, st));
#else
        return MOGE_ERR_INVALID;                     // the fused residual block lives in tools/experiments/ (experiments builds only)
#endif
    } else {
        TL(launch_gemm<T>(g, AMODE_CONV3, st));
    }
    if (a.dot_w) {
        if (!conv_pp_eligible(g)) return MOGE_ERR_INVALID;
        TCHK(hipMemcpyAsync(a.y, dout.p, (size_t)B * Ho * Wo * 4 * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        TL(from_t<T>(yb.p, a.y, (long)ny, st));
    }
    TCHK(hipStreamSynchronize(st));
    return 0;
}

template <typename T>
static int t_convt(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, hipStream_t st) {
    DevBuf xb, wb, yb, bb;
    const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * 4 * H * W * Cout;
    TCHK(xb.alloc(nx * sizeof(T))); TCHK(wb.alloc((size_t)4 * Cout * Cin * sizeof(T))); TCHK(yb.alloc(ny * sizeof(T))); TCHK(bb.alloc(4 * Cout * sizeof(float)));
    TL(to_t<T>(x, xb.p, (long)nx, st));
    TL(launch_repack<T>(w, wb.p, 4, Cout, 1, Cin, 1, 4, 0, (long)Cout * 4, (long)Cout * Cin, Cin, 0, st));
    TL(launch_repack<float>(bias, bb.p, 4, 1, 1, Cout, 0, 0, 0, 1, Cout, 0, 0, st));
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.a = xb.p; g.lda = Cin; g.w = wb.p; g.ldw = Cin; g.M = B * H * W; g.N = 4 * Cout; g.K = Cin;
    g.epi = EPI_CONVT; g.bias = (const float*)bb.p; g.out = yb.p; g.Cout = Cout; g.pixW = W; g.pixH = H;
    TL(launch_gemm<T>(g, AMODE_LINEAR, st));
    TL(from_t<T>(yb.p, y, (long)ny, st));
    TCHK(hipStreamSynchronize(st));
    return 0;
}

template <typename T>
static int t_norm_act(const float* x, const float* gamma, const float* beta, float* y, int B, int H, int W, int C, int G, int act, int in_place, hipStream_t st) {
    const size_t n = (size_t)B * H * W * C;
    DevBuf xb, yb, sc;
    TCHK(xb.alloc(n * sizeof(T))); TCHK(yb.alloc(n * sizeof(T))); TCHK(sc.alloc((groupnorm_scratch_floats(B, H, W, G > 0 ? G : 1) + 4) * sizeof(float)));
    TL(to_t<T>(x, xb.p, (long)n, st));
    TL(launch_groupnorm_act<T>(xb.p, in_place ? xb.p : yb.p, gamma, beta, (float*)sc.p, B, H, W, C, G, act, st));
    TL(from_t<T>(in_place ? xb.p : yb.p, y, (long)n, st));
    TCHK(hipStreamSynchronize(st));
    return 0;
}
template <typename T>
static int t_groupnorm(const float* x, const float* gamma, const float* beta, float* y, int B, int H, int W, int C, int G, hipStream_t st) {
    const size_t n = (size_t)B * H * W * C;
    DevBuf xb, yb, sc;
    TCHK(xb.alloc(n * sizeof(T))); TCHK(yb.alloc(n * sizeof(T))); TCHK(sc.alloc(groupnorm_scratch_floats(B, H, W, G) * sizeof(float)));
    TL(to_t<T>(x, xb.p, (long)n, st));
    TL(launch_groupnorm_relu<T>(xb.p, yb.p, gamma, beta, (float*)sc.p, B, H, W, C, G, st));
    TL(from_t<T>(yb.p, y, (long)n, st));
    TCHK(hipStreamSynchronize(st));
    return 0;
}

// fused ConvTranspose2d + 3x3 (conv_pp.hip CT3 + border), fp16 only: x (B,H,W,Cin), wt (Cin,Cout,2,2), bt (Cout), w3 (Cout,Cout,3,3), b3 (Cout), optional
// side (B,2H,2W,Cout) with side_w (Cout,Cout), optional uv at the output resolution -> y (B,2H,2W,Cout)
static int t_ct3(const moge_test_ct3_args& a, hipStream_t st) {
    const int B = a.B, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * 4 * H * W * Cout;
    DevBuf xb, yb, wc, dw, b4, sb, sw;
    TCHK(xb.alloc(nx * 2)); TCHK(yb.alloc(ny * 2)); TCHK(wc.alloc((size_t)16 * Cout * Cin * 2)); TCHK(dw.alloc((size_t)24 * Cout * Cin * 2)); TCHK(b4.alloc((size_t)4 * Cout * 4));
    TL(to_t<f16>(a.x, xb.p, (long)nx, st));
    TL(ct3_compose_device(a.w3, a.wt, Cin, Cout, wc.p, dw.p, st));
    std::vector<float> w3((size_t)Cout * Cout * 9), bt(Cout), b3(Cout), bias4((size_t)4 * Cout);
    TCHK(hipMemcpyAsync(w3.data(), a.w3, w3.size() * 4, hipMemcpyDeviceToHost, st));
    TCHK(hipMemcpyAsync(bt.data(), a.bt, (size_t)Cout * 4, hipMemcpyDeviceToHost, st));
    TCHK(hipMemcpyAsync(b3.data(), a.b3, (size_t)Cout * 4, hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    for (int o = 0; o < Cout; o++) {
        double acc = b3[o];
        for (int m = 0; m < Cout; m++) {
            double t9 = 0;
            for (int k = 0; k < 9; k++) t9 += w3[((size_t)o * Cout + m) * 9 + k];
            acc += t9 * bt[m];
        }
        for (int q = 0; q < 4; q++) bias4[(size_t)q * Cout + o] = (float)acc;
    }
    TCHK(hipMemcpyAsync(b4.p, bias4.data(), bias4.size() * 4, hipMemcpyHostToDevice, st));
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.a = xb.p; g.H = H; g.W = W; g.C = Cin; g.w = wc.p; g.ldw = 4 * Cin; g.M = B * H * W; g.N = 4 * Cout; g.K = 4 * Cin;
    g.epi = EPI_CONVT; g.bias = (const float*)b4.p; g.out = yb.p; g.Cout = Cout; g.pixW = W; g.pixH = H; g.ct3 = 1;
    if (a.side) {
        TCHK(sb.alloc(ny * 2)); TCHK(sw.alloc((size_t)Cout * Cout * 2));
        TL(to_t<f16>(a.side, sb.p, (long)ny, st));
        TL(to_t<f16>(a.side_w, sw.p, (long)Cout * Cout, st));
        g.a2 = sb.p; g.w2 = sw.p;
    }
    if (a.wu) { g.uv.wu = a.wu; g.uv.wv = a.wv; g.uv.u0 = a.u0; g.uv.u1 = a.u1; g.uv.v0 = a.v0; g.uv.v1 = a.v1;
                g.uv.ustep = 2 * W > 1 ? (a.u1 - a.u0) / (float)(2 * W - 1) : 0.f; g.uv.vstep = 2 * H > 1 ? (a.v1 - a.v0) / (float)(2 * H - 1) : 0.f; }
    if (!conv_pp_eligible(g)) return MOGE_ERR_INVALID;
    TL(launch_conv_pp(g, st));
    if (!a.no_border) TL(launch_ct3_border(xb.p, dw.p, yb.p, B, H, W, Cin, Cout, st));
    TL(from_t<f16>(yb.p, a.y, (long)ny, st));
    TCHK(hipStreamSynchronize(st));
    return 0;
}

extern "C" {

int moge_test_ct3(const moge_test_ct3_args* args, void* stream) {
    if (!args || args->precision != MOGE_FP16) return MOGE_ERR_INVALID;
    return t_ct3(*args, (hipStream_t)stream);
}

int moge_test_gemm(int precision, const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int act, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    return precision == MOGE_FP16 ? t_gemm<f16>(A, W, bias, C, M, N, K, act, st) : t_gemm<float>(A, W, bias, C, M, N, K, act, st);
}

int moge_test_gemm_ex(const moge_test_gemm_args* args, void* stream) {
    if (!args) return MOGE_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    return args->precision == MOGE_FP16 ? t_gemm_ex<f16>(*args, st) : t_gemm_ex<float>(*args, st);
}

int moge_test_layernorm(int precision, const float* x, const float* w, const float* b, float* y, int rows, int D, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (precision == MOGE_FP16) {
        DevBuf yb;
        TCHK(yb.alloc((size_t)rows * D * sizeof(f16)));
        TL(launch_layernorm<f16>(x, w, b, yb.p, nullptr, rows, D, D, 0, 0, 1, st));
        TL((launch_convert<f16, float>(yb.p, y, (long)rows * D, st)));
    } else {
        TL(launch_layernorm<float>(x, w, b, y, nullptr, rows, D, D, 0, 0, 1, st));
    }
    TCHK(hipStreamSynchronize(st));
    return 0;
}

int moge_test_attention(int precision, const float* q, const float* k, const float* v, float* o, int B, int nh, int N, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    return precision == MOGE_FP16 ? t_attention<f16>(q, k, v, o, B, nh, N, st) : t_attention<float>(q, k, v, o, B, nh, N, st);
}

int moge_test_conv3x3(int precision, const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, int relu_in,
                      void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int up2 = (relu_in >> 1) & 1, relu = relu_in & 1;      // bit 1 of relu_in selects bilinear x2 + 3x3 (4-phase conv)
    return precision == MOGE_FP16 ? t_conv3<f16>(x, w, bias, y, B, H, W, Cin, Cout, relu, up2, st) : t_conv3<float>(x, w, bias, y, B, H, W, Cin, Cout, relu, up2, st);
}

int moge_test_conv_ex(const moge_test_conv_args* args, void* stream) {
    if (!args) return MOGE_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    return args->precision == MOGE_FP16 ? t_conv_ex<f16>(*args, st) : t_conv_ex<float>(*args, st);
}

int moge_test_convt2x2(int precision, const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    return precision == MOGE_FP16 ? t_convt<f16>(x, w, bias, y, B, H, W, Cin, Cout, st) : t_convt<float>(x, w, bias, y, B, H, W, Cin, Cout, st);
}

int moge_test_preprocess(const float* image, float* out, int B, int H, int W, int rows, int cols, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
    TL((launch_preprocess<float, float>(image, out, B, H, W, rows, cols, 0, 1, 0, 1, mean, sd, st)));
    TCHK(hipStreamSynchronize(st));
    return 0;
}

int moge_test_resize_bicubic_aa(const float* image, float* out, int B, int H, int W, int OH, int OW, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    TL(launch_resize_bicubic_aa<float>(image, out, B, H, W, OH, OW, 0, st));
    TCHK(hipStreamSynchronize(st));
    return 0;
}

int moge_test_norm_act(int precision, const float* x, const float* gamma, const float* beta, float* y, int B, int H, int W, int C, int groups, int act, int in_place, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    return precision == MOGE_FP16 ? t_norm_act<f16>(x, gamma, beta, y, B, H, W, C, groups, act, in_place, st) : t_norm_act<float>(x, gamma, beta, y, B, H, W, C, groups, act, in_place, st);
}

int moge_test_groupnorm_relu(int precision, const float* x, const float* gamma, const float* beta, float* y, int B, int H, int W, int C, int groups, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    return precision == MOGE_FP16 ? t_groupnorm<f16>(x, gamma, beta, y, B, H, W, C, groups, st) : t_groupnorm<float>(x, gamma, beta, y, B, H, W, C, groups, st);
}

int moge_test_posembed(const float* pos, float* out, int D, int rows, int cols, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    TL(launch_posembed(pos, out, D, rows, cols, 0, st));
    TCHK(hipStreamSynchronize(st));
    return 0;
}

int moge_test_recover(const float* points, const uint8_t* mask, const float* focal_in, int B, int H, int W, float* focal, float* shift,
                      int32_t* status, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    TL(launch_recover(points, nullptr, mask, nullptr, focal_in, B, H, W, focal, shift, nullptr, status, st));
    TCHK(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
