// MFMA GEMM / implicit-GEMM convolution for gfx950 with fused epilogues.
//
//   C[M,N] = A[M,K] * W[N,K]^T      A: activations (token-major / NHWC), W: packed weights, both K-contiguous.
//
// One kernel template serves every matrix-shaped op of the MoGe-2 hot path:
//   linear layers of the ViT        attention.py:72,79  mlp.py:35,38  patch_embed.py:75 (conv14/s14 as GEMM)
//   1x1 convs                       modules.py:128-131, 209, 231
//   ConvTranspose2d k2/s2           modules.py:162  (GEMM to 4*Cout + pixel-shuffle store)
//   3x3 replicate-padded convs      modules.py:53,59,148-181 (implicit GEMM, K = 9*Cin, clamp-indexed loads,
//                                   optional ReLU prologue); bilinear x2 + 3x3 (modules.py:157-158) runs as a 4-phase
//                                   3x3 conv on the LOW-res map with pre-combined weights (N = 4*Cout) + pixel shuffle
//
// Design (CDNA4): the block computes a BM x BN tile; K is consumed in 128-byte slabs (64 halves / 32 floats) that
// are staged global -> registers -> LDS (16-byte chunks, XOR-swizzled so ds_read_b128 is bank-conflict free),
// double buffered with one barrier per slab and the next slab's global loads in flight during the MFMAs.
// The MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows) so each lane owns one
// output row m and 4 consecutive output columns n per register quad -> vector stores and vector bias loads.
// Storage type T selects the instruction: f16 -> v_mfma_f32_32x32x16_f16, float -> v_mfma_f32_32x32x2_f32
// (exact fp32, the parity mode).  Accumulation is fp32 in both.
#include "common.h"
#include <cstdlib>

template <typename T>
__device__ __forceinline__ void epilogue4(const GemmArgs& g, int m, int n, float v0, float v1, float v2, float v3) {
    if (m >= g.M || n >= g.N) return;
    float v[4] = {v0, v1, v2, v3};
    if (g.epi == EPI_RESID && !g.xres) {
        // fp16 residual stream (`.half()` models; gemm_pp.hip EPK_RESID16): x16 <- fp16(fp32(x16) + gamma (acc + bias)), no statistics on this path
        f16* p = reinterpret_cast<f16*>(g.x16) + (size_t)m * g.ldc + n;
        const f16x4 h = *reinterpret_cast<const f16x4*>(p);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(g.gamma + n);
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) b = *reinterpret_cast<const f32x4*>(g.bias + n);
        f16x4 o;
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = (f16)((float)h[i] + resid_term(gm[i], v[i], b[i]));
        *reinterpret_cast<f16x4*>(p) = o;
        return;
    }
    if (g.epi == EPI_RESID) {
        // x[m][n] += gamma[n] * (acc + bias[n]); same operation order as gemm_pp.hip (resid_term, then one add)
        float* p = g.xres + (size_t)m * g.ldc + n;
        f32x4 x = *reinterpret_cast<f32x4*>(p);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(g.gamma + n);
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) b = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = x[i] + resid_term(gm[i], v[i], b[i]);
        *reinterpret_cast<f32x4*>(p) = x;
        return;
    }
    if (g.ln_mr) {           // LN-fold consumer (same arithmetic as gemm_pp.hip: ln_fold_term)
        const f32x2 mr = *reinterpret_cast<const f32x2*>(g.ln_mr + 2 * (size_t)m);
        const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n);
        const f32x4 lc = *reinterpret_cast<const f32x4*>(g.ln_c + n);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = ln_fold_term(v[i], mr[0], mr[1], lc[i], b[i]);
    } else if (g.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] += b[i];
    }
    switch (g.epi) {
    case EPI_STORE: {
        if (g.uv.wu) {
            const int x = m % g.pixW;
            const int y = (m / g.pixW) % g.pixH;
            const float u = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, g.pixW, x);
            const float vv = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, g.pixH, y);
            const f32x4 wu = *reinterpret_cast<const f32x4*>(g.uv.wu + n);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(g.uv.wv + n);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = uv_term_add(v[i], wu[i], u, wv[i], vv);
        }
        if (g.add) {
            float a[4];
            load4(reinterpret_cast<const T*>(g.add) + (size_t)m * g.ldadd + n, a);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] += a[i];
        }
        if (g.act == ACT_RELU) {
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.f);
        } else if (g.act == ACT_GELU) {
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = TT<T>::PREC ? gelu_fast(v[i]) : gelu_erf(v[i]);   // fp16 storage: the fast form (as gemm_pp.hip)
        }
        store4(reinterpret_cast<T*>(g.out) + (size_t)m * g.ldc + n, v[0], v[1], v[2], v[3]);
        break;
    }
    case EPI_PATCH: {
        const int b = m / g.Np, p = m - b * g.Np;
        const f32x4 pe = *reinterpret_cast<const f32x4*>(g.pos + (size_t)(1 + p) * g.N + n);
        f32x4 o = {v[0] + pe[0], v[1] + pe[1], v[2] + pe[2], v[3] + pe[3]};
        *reinterpret_cast<f32x4*>(g.xres + ((size_t)b * g.Ntok + 1 + p) * g.N + n) = o;
        if (p == 0 && g.cls) {            // the image's cls row (vision_transformer.py:228-231: cls_token + pos_embed[0]); cls_row_kernel's arithmetic, one launch fewer
            const f32x4 c = *reinterpret_cast<const f32x4*>(g.cls + n), p0 = *reinterpret_cast<const f32x4*>(g.pos + n);
            *reinterpret_cast<f32x4*>(g.xres + (size_t)b * g.Ntok * g.N + n) = f32x4{c[0] + p0[0], c[1] + p0[1], c[2] + p0[2], c[3] + p0[3]};
        }
        break;
    }
    case EPI_QKV: {
        const int which = n / g.D;
        const int rem = n - which * g.D;
        const int hd = rem >> 6, d = rem & 63;
        const int b = m / g.Ntok, tok = m - b * g.Ntok;
        const size_t bh = (size_t)b * g.nh + hd;
        if (which == 0) {
            store4(reinterpret_cast<T*>(g.q) + (bh * g.Ntok + tok) * 64 + d, q_scaled(v[0], g.qscale), q_scaled(v[1], g.qscale), q_scaled(v[2], g.qscale), q_scaled(v[3], g.qscale));
        } else if (which == 1) {
            store4(reinterpret_cast<T*>(g.k) + (bh * g.Ntok + tok) * 64 + d, v[0], v[1], v[2], v[3]);
        } else if (g.v_rowmajor) {
            store4(reinterpret_cast<T*>(g.vT) + (bh * g.Ntok + tok) * 64 + d, v[0], v[1], v[2], v[3]);
        } else {
            T* p = reinterpret_cast<T*>(g.vT) + (bh * 64 + d) * (size_t)g.Npad + tok;
#pragma unroll
            for (int i = 0; i < 4; i++) p[(size_t)i * g.Npad] = (T)v[i];
        }
        break;
    }
    case EPI_CONVT: {
        const int qd = n / g.Cout, co = n - qd * g.Cout;
        const int dy = qd >> 1, dx = qd & 1;
        const int x = m % g.pixW;
        const int t = m / g.pixW;
        const int y = t % g.pixH, b = t / g.pixH;
        if (g.uv.wu && g.uv_in) {          // uv channels of the ConvTranspose2d INPUT: low-res pixel (y, x), weights per column n = (dy, dx, co)
            const float u = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, g.pixW, x);
            const float vv = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, g.pixH, y);
            const f32x4 wu = *reinterpret_cast<const f32x4*>(g.uv.wu + n);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(g.uv.wv + n);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] += wu[i] * u + wv[i] * vv;
        } else if (g.uv.wu) {          // uv term evaluated at the HIGH-res pixel (2y+dy, 2x+dx); wu/wv indexed by output channel
            const float u = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, 2 * g.pixW, 2 * x + dx);
            const float vv = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, 2 * g.pixH, 2 * y + dy);
            const f32x4 wu = *reinterpret_cast<const f32x4*>(g.uv.wu + co);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(g.uv.wv + co);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] += wu[i] * u + wv[i] * vv;
        }
        const size_t idx = (((size_t)b * 2 * g.pixH + 2 * y + dy) * (2 * g.pixW) + 2 * x + dx) * g.Cout + co;
        store4(reinterpret_cast<T*>(g.out) + idx, v[0], v[1], v[2], v[3]);
        break;
    }
    }
}

// fp16 residual stream, one quad: x16[m][n..n+3] <- fp16(fp32(x16) + gamma (acc + bias)); returns the ROUNDED values (what the LN-fold statistics
// are taken from in this mode, as gemm_pp.hip's pp_resid16_store does)
__device__ __forceinline__ f32x4 resid16_quad(const GemmArgs& g, int m, int n, const float* a) {
    f16* p = reinterpret_cast<f16*>(g.x16) + (size_t)m * g.ldc + n;
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(g.gamma + n);
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) b = *reinterpret_cast<const f32x4*>(g.bias + n);
    f16x4 o;
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = (f16)((float)h[i] + resid_term(gm[i], a[i], b[i]));
    *reinterpret_cast<f16x4*>(p) = o;
    return f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
}

// One 32 x 32 accumulator tile of a lane: row m, quads at columns nb + 8q (nb already includes 4*hi).  EPI_RESID with the LN-fold
// producer outputs (GemmArgs::x16 / ln_part): residual update + fp16 copy + the (sum, sum of squares) of the row's 32-column group with the
// SAME summation tree as gemm_pp.hip's pp_resid_rows (quad sums, then pairs 2q/2q+1 across the two half-waves, then (0+1)+(2+3)), so the
// folded LayerNorm statistics - and with them every output - do not depend on which kernel a batch size selects.
template <typename T>
__device__ __forceinline__ void epilogue_tile32(const GemmArgs& g, int m, int nb, int hi, const f32x16& a) {
    if (g.epi == EPI_RESID && g.x16) {
        const bool ok = m < g.M;
        float t1[4], t2[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n = nb + 8 * q;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (ok && !g.xres) {
                const float aq[4] = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
                x = resid16_quad(g, m, n, aq);
            } else if (ok) {
                float* p = g.xres + (size_t)m * g.ldc + n;
                x = *reinterpret_cast<f32x4*>(p);
                const f32x4 gm = *reinterpret_cast<const f32x4*>(g.gamma + n);
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (g.bias) b = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = x[i] + resid_term(gm[i], a[4 * q + i], b[i]);
                *reinterpret_cast<f32x4*>(p) = x;
                *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(g.x16) + (size_t)m * g.ldc + n) = f16x4{(f16)x[0], (f16)x[1], (f16)x[2], (f16)x[3]};
            }
            float s1, s2;
            ln_quad_sums(x, s1, s2);
            t1[q] = s1 + __shfl_xor(s1, 32);
            t2[q] = s2 + __shfl_xor(s2, 32);
        }
        const float u1 = (t1[0] + t1[1]) + (t1[2] + t1[3]), u2 = (t2[0] + t2[1]) + (t2[2] + t2[3]);
        if (hi == 0 && ok && g.ln_part) *reinterpret_cast<f32x2*>(g.ln_part + ((size_t)m * (g.N >> 5) + (nb >> 5)) * 2) = f32x2{u1, u2};
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) epilogue4<T>(g, m, nb + 8 * q, a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}

// 16x16x32 accumulators (fp16 LINEAR kernels): two tiles of the same 16 rows covering one 32-column group.  Lane: row m = .. + (lane & 15),
// columns nb32 + jh*16 + 4*(lane >> 4) .. +3.  The LN-fold producer's summation tree is the one of gemm_pp.hip / epilogue_tile32: quad index
// c = jh*4 + (lane >> 4); pairs (c, c^1) across lanes +-16, (c, c^2) across lanes +-32, then the two tiles.
template <typename T>
__device__ __forceinline__ void epilogue_pair16(const GemmArgs& g, int m, int nb32, int g4, const f32x4& a0, const f32x4& a1) {
    if (g.epi == EPI_RESID && g.x16) {
        const bool ok = m < g.M;
        float u1[2], u2[2];
#pragma unroll
        for (int jh = 0; jh < 2; jh++) {
            const f32x4& a = jh ? a1 : a0;
            const int n = nb32 + jh * 16 + 4 * g4;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (ok && !g.xres) {
                const float aq[4] = {a[0], a[1], a[2], a[3]};
                x = resid16_quad(g, m, n, aq);
            } else if (ok) {
                float* p = g.xres + (size_t)m * g.ldc + n;
                x = *reinterpret_cast<f32x4*>(p);
                const f32x4 gm = *reinterpret_cast<const f32x4*>(g.gamma + n);
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (g.bias) b = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = x[i] + resid_term(gm[i], a[i], b[i]);
                *reinterpret_cast<f32x4*>(p) = x;
                *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(g.x16) + (size_t)m * g.ldc + n) = f16x4{(f16)x[0], (f16)x[1], (f16)x[2], (f16)x[3]};
            }
            float s1, s2;
            ln_quad_sums(x, s1, s2);
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            u1[jh] = s1; u2[jh] = s2;
        }
        // (write-through: with GemmArgs::ln_mr_out another workgroup of THIS launch, possibly on another XCD's L2, reads the partials - see the fused finalize)
        if (g4 == 0 && ok && g.ln_part)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(g.ln_part + ((size_t)m * (g.N >> 5) + (nb32 >> 5)) * 2),
                               __builtin_bit_cast(unsigned long long, f32x2{u1[0] + u1[1], u2[0] + u2[1]}), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    epilogue4<T>(g, m, nb32 + 4 * g4, a0[0], a0[1], a0[2], a0[3]);
    epilogue4<T>(g, m, nb32 + 16 + 4 * g4, a1[0], a1[1], a1[2], a1[3]);
}

template <typename T, int WM, int WN, int TM, int TN, int AMODE>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmArgs g) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int CH = TT<T>::CH;
    constexpr int A_IT = BM * 8 / NT, W_IT = BN * 8 / NT;
    constexpr int RSTEP = NT / 8;                      // tile rows covered per load pass
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile/thread mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                    // [2][BM][128 B]
    char* sW = smem + 2 * BM * 128;     // [2][BN][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int nbn = (g.N + BN - 1) / BN;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x - bm * nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int Kchunks = g.K / CH;
    const int nkt = (Kchunks + 7) >> 3;
    const int c = tid & 7;              // this thread's chunk column inside every 128-byte slab
    const int r0 = tid >> 3;            // first tile row this thread stages

    // ---- per-thread A row descriptors ------------------------------------------------------
    const T* a_base[A_IT];              // LINEAR: row pointer; CONV: image base pointer of the row's batch item
    int a_y[A_IT], a_x[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
        int m = m0 + r0 + i * RSTEP;
        m = m < g.M ? m : g.M - 1;
        if (AMODE == AMODE_LINEAR) {
            a_base[i] = reinterpret_cast<const T*>(g.a) + (size_t)m * g.lda;
            a_y[i] = 0; a_x[i] = 0;
        } else {
            const int x = m % g.W;
            const int t = m / g.W;
            const int y = t % g.H, b = t / g.H;
            a_base[i] = reinterpret_cast<const T*>(g.a) + (size_t)b * g.H * g.W * g.C;
            a_y[i] = y; a_x[i] = x;
        }
    }
    const T* w_base[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; i++) {
        int n = n0 + r0 + i * RSTEP;
        n = n < g.N ? n : g.N - 1;
        w_base[i] = reinterpret_cast<const T*>(g.w) + (size_t)n * g.ldw;
    }
    const int cpc = (AMODE == AMODE_LINEAR) ? 1 : g.C / CH;     // chunks per conv tap

    u32x4 ra[A_IT], rw[W_IT];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_slab = [&](int kt) {
        const int kc = kt * 8 + c;
        const bool kvalid = kc < Kchunks;
        if (AMODE == AMODE_LINEAR) {
#pragma unroll
            for (int i = 0; i < A_IT; i++)
                ra[i] = kvalid ? *reinterpret_cast<const u32x4*>(a_base[i] + (size_t)kc * CH) : zero4;
        } else {
            const int tap = kc / cpc;
            const int cc = kc - tap * cpc;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
            for (int i = 0; i < A_IT; i++) {
                int yy = a_y[i] + dy, xx = a_x[i] + dx;
                yy = yy < 0 ? 0 : (yy > g.H - 1 ? g.H - 1 : yy);      // replicate padding (modules.py:53)
                xx = xx < 0 ? 0 : (xx > g.W - 1 ? g.W - 1 : xx);
                ra[i] = kvalid ? *reinterpret_cast<const u32x4*>(a_base[i] + ((size_t)yy * g.W + xx) * g.C + cc * CH) : zero4;
            }
        }
#pragma unroll
        for (int i = 0; i < W_IT; i++)
            rw[i] = kvalid ? *reinterpret_cast<const u32x4*>(w_base[i] + (size_t)kc * CH) : zero4;
    };

    auto store_slab = [&](int buf) {
        char* dA = sA + buf * BM * 128;
        char* dW = sW + buf * BN * 128;
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            const int row = r0 + i * RSTEP;
            u32x4 v = ra[i];
            if (g.relu_in) v = relu_chunk<T>(v);
            *reinterpret_cast<u32x4*>(dA + row * 128 + (swz<8>(row, c) << 4)) = v;
        }
#pragma unroll
        for (int i = 0; i < W_IT; i++) {
            const int row = r0 + i * RSTEP;
            *reinterpret_cast<u32x4*>(dW + row * 128 + (swz<8>(row, c) << 4)) = rw[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    load_slab(0);
    store_slab(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_slab(kt + 1);            // global loads in flight during the MFMAs below
        const char* cA = sA + buf * BM * 128;
        const char* cW = sW + buf * BN * 128;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int cs = 2 * s + hi;
            u32x4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int row = (wm * TM + i) * 32 + l31;
                af[i] = *reinterpret_cast<const u32x4*>(cA + row * 128 + (swz<8>(row, cs) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int row = (wn * TN + j) * 32 + l31;
                wf[j] = *reinterpret_cast<const u32x4*>(cW + row * 128 + (swz<8>(row, cs) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) mma_step<T>(acc[i][j], wf[j], af[i]);     // D[n][m]
        }
        if (kt + 1 < nkt) store_slab(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns row m, register quad g -> columns n..n+3 -------------------------
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = m0 + (wm * TM + i) * 32 + l31;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nb = n0 + (wn * TN + j) * 32 + 4 * hi;
            epilogue_tile32<T>(g, m, nb, hi, acc[i][j]);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// LINEAR-mode variant with direct-to-LDS staging (global_load_lds_dwordx4): no VGPR round trip, no ds_write pass.
// One wave-instruction fills 8 tile rows (1 KiB, lane-linear in LDS); the XOR swizzle is applied to the per-lane
// SOURCE chunk (lane p of a row fetches logical chunk p ^ ((row>>1)&7)), reads use the same XOR.  Requirements:
// K % (8*CH) == 0 (no K tail: the DMA cannot zero-fill), no ReLU prologue.  Blocks are remapped so that each XCD
// owns a contiguous range of tiles (A row-panels stay in one XCD's L2).
// ------------------------------------------------------------------------------------------------------------
#define GLDS_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define GLDS_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <typename T, int WM, int WN, int TM, int TN, int NSTAGE>
__global__ __launch_bounds__(64 * WM * WN) void gemm_glds_kernel(const GemmArgs g) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int CH = TT<T>::CH;
    constexpr int RSTEP = NT / 8;
    constexpr int A_IT = BM / RSTEP, W_IT = BN / RSTEP;
    static_assert(BM % RSTEP == 0 && BN % RSTEP == 0 && (RSTEP % 16) == 0, "tile/thread mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sW = smem + NSTAGE * BM * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int nbn = (g.N + BN - 1) / BN;
    // XCD-aware bijective remap (block b runs on XCD b % 8): XCD x gets a contiguous chunk of tile ids
    const int nwg = gridDim.x;
    int wg;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int bm = wg / nbn, bn = wg - bm * nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int nkt = g.K / (8 * CH);
    const int r0 = tid >> 3;
    const int csrc = (tid & 7) ^ ((r0 >> 1) & 7);       // logical chunk this lane fetches (same for every pass: RSTEP % 16 == 0)

    // DMA sources as  tile base (wave-uniform, SGPR pair) + per-lane 32-bit byte offset  (< BM x lda x 4 bytes): with the offset made opaque at the
    // point of use the compiler selects  global_load_lds_dwordx4 vOff, s[base:base+1]  - no per-lane 64-bit pointers (2 VGPRs + one 64-bit VALU add
    // per piece and K-step) in the ring loop of the latency-regime kernels.
    unsigned a_off[A_IT], w_off[W_IT];
    const char* a_tile = reinterpret_cast<const char*>(reinterpret_cast<const T*>(g.a) + (size_t)m0 * g.lda);
    const char* w_tile = reinterpret_cast<const char*>(reinterpret_cast<const T*>(g.w) + (size_t)n0 * g.ldw);
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
        int m = m0 + r0 + i * RSTEP;
        m = m < g.M ? m : g.M - 1;
        a_off[i] = (unsigned)(((m - m0) * g.lda + csrc * CH) * (int)sizeof(T));
    }
#pragma unroll
    for (int i = 0; i < W_IT; i++) {
        int n = n0 + r0 + i * RSTEP;
        n = n < g.N ? n : g.N - 1;
        w_off[i] = (unsigned)(((n - n0) * g.ldw + csrc * CH) * (int)sizeof(T));
    }
    const int wrow = wave * 8;                           // first tile row this wave fills in each pass

    auto issue = [&](int kt, int buf) {
        char* dA = sA + buf * BM * 128 + wrow * 128;
        char* dW = sW + buf * BN * 128 + wrow * 128;
        const char* ak = uniform_ptr(a_tile + (size_t)kt * 8 * CH * sizeof(T));
        const char* wk = uniform_ptr(w_tile + (size_t)kt * 8 * CH * sizeof(T));
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            unsigned o = a_off[i];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds(GLDS_GPTR(ak + o), GLDS_LPTR(dA + i * RSTEP * 128), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < W_IT; i++) {
            unsigned o = w_off[i];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds(GLDS_GPTR(wk + o), GLDS_LPTR(dW + i * RSTEP * 128), 16, 0, 0);
        }
    };

    // fp16: v_mfma_f32_16x16x32_f16 (the same instruction and K grouping as gemm_pp128m16: a GEMM's result does not depend on which of the
    // two kernels its batch size selects); fp32: v_mfma_f32_32x32x2_f32
    constexpr bool M16 = TT<T>::PREC == 1;
    f32x16 acc[TM][TN];
    f32x4 acc16[2 * TM][2 * TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * TM; i++)
#pragma unroll
        for (int j = 0; j < 2 * TN; j++) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, g4 = lane >> 4;

    auto compute = [&](int buf, auto&& mid) {             // mid(): issued between the fragment reads and the MFMAs (the ring form's DMA requests)
        const char* cA = sA + buf * BM * 128;
        const char* cW = sW + buf * BN * 128;
        if constexpr (M16) {
            // NSTAGE >= 3 (the counted-wait ring; the latency-regime dispatch): the fragments of BOTH K-steps are requested before the first MFMA.
            // The compiler waits lgkmcnt(0) in front of each K-step's MFMAs whatever is in flight (it does not count LDS reads in these kernels),
            // so read / wait / MFMAs twice per slab exposed the LDS latency twice.  Same MFMAs in the same order (bit-identical).  Old / new
            // library alternating on one box (profiles/r04zd_kbench_gemm_lat_ab.log): fc2 of one image 628 -> 675 TF/s, proj 409 -> 426,
            // ViT-B fc2 490 -> 517.  The one- and two-stage forms (__syncthreads() per slab) measured 3-12 % SLOWER that way and keep the
            // per-K-step order.
            constexpr int NKS = NSTAGE >= 3 ? 2 : 1;              // K-steps per request group
            u32x4 af[NKS][2 * TM], wf[NKS][2 * TN];
#pragma unroll
            for (int k0 = 0; k0 < 2; k0 += NKS) {
#pragma unroll
                for (int kk = 0; kk < NKS; kk++) {
                    const int cs = 4 * (k0 + kk) + g4;
#pragma unroll
                    for (int i = 0; i < 2 * TM; i++) {
                        const int row = (wm * 2 * TM + i) * 16 + l15;
                        af[kk][i] = *reinterpret_cast<const u32x4*>(cA + row * 128 + (swz<8>(row, cs) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < 2 * TN; j++) {
                        const int row = (wn * 2 * TN + j) * 16 + l15;
                        wf[kk][j] = *reinterpret_cast<const u32x4*>(cW + row * 128 + (swz<8>(row, cs) << 4));
                    }
                }
                if constexpr (NKS == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    mid();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int kk = 0; kk < NKS; kk++)
#pragma unroll
                    for (int i = 0; i < 2 * TM; i++)
#pragma unroll
                        for (int j = 0; j < 2 * TN; j++) mma16<f16>(acc16[i][j], wf[kk][j], af[kk][i]);
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int cs = 2 * s + hi;
            u32x4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int row = (wm * TM + i) * 32 + l31;
                af[i] = *reinterpret_cast<const u32x4*>(cA + row * 128 + (swz<8>(row, cs) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int row = (wn * TN + j) * 32 + l31;
                wf[j] = *reinterpret_cast<const u32x4*>(cW + row * 128 + (swz<8>(row, cs) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) mma_step<T>(acc[i][j], wf[j], af[i]);
        }
    };

    if constexpr (NSTAGE == 1) {
        // single LDS buffer, two barriers per slab: latency is hidden by co-resident blocks (32 KiB LDS -> 3-4 blocks/CU)
        for (int kt = 0; kt < nkt; kt++) {
            issue(kt, 0);
            __syncthreads();
            compute(0, [] {});
            __syncthreads();
        }
    } else if constexpr (NSTAGE == 2) {
        issue(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nkt; kt++) {
            const int buf = kt & 1;
            if (kt + 1 < nkt) issue(kt + 1, buf ^ 1);
            compute(buf, [] {});
            __syncthreads();          // drains the DMA (vmcnt(0)) and orders it against the next slab's reads
        }
    } else {
        // NSTAGE-slab ring, DMA NSTAGE - 1 slabs ahead, ONE raw barrier per slab and a COUNTED vmcnt so the younger slabs' loads stay in
        // flight across the barrier (a __syncthreads() here would drain them: LDS-DMA counts as a pending LDS write).
        constexpr int G = A_IT + W_IT;           // DMA instructions per thread per slab
        constexpr int AHEAD = NSTAGE - 1;
        static_assert(G * (AHEAD - 1) <= 63, "vmcnt field");
#pragma unroll
        for (int s = 0; s < AHEAD; s++)
            if (s < nkt) issue(s, s);
        int cur = 0, nxt = AHEAD;                // slab kt lives in buffer kt % NSTAGE
        for (int kt = 0; kt < nkt; kt++) {
            const int younger = nkt - 1 - kt;    // slabs issued after slab kt (capped at AHEAD - 1)
            if (younger >= AHEAD - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (AHEAD - 1)) : "memory");
            else if (AHEAD > 2 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * 2) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();        // slab kt landed for every wave; every wave is done reading slab kt-1
            // the slab's DMA requests go BEHIND its fragment reads (fp16 form): a piece costs tens of issue clocks, and in front of the reads that
            // time sat between the barrier and the first MFMA of every slab
            if constexpr (M16) {
                compute(cur, [&] { if (kt + AHEAD < nkt) issue(kt + AHEAD, nxt); });
            } else {
                if (kt + AHEAD < nkt) issue(kt + AHEAD, nxt);
                compute(cur, [] {});
            }
            cur = cur == NSTAGE - 1 ? 0 : cur + 1;
            nxt = nxt == NSTAGE - 1 ? 0 : nxt + 1;
        }
    }
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < 2 * TM; i++) {
            const int m = m0 + (wm * 2 * TM + i) * 16 + l15;
#pragma unroll
            for (int j = 0; j < TN; j++) epilogue_pair16<T>(g, m, n0 + (wn * TN + j) * 32, g4, acc16[i][2 * j], acc16[i][2 * j + 1]);
        }
        if (g.ln_mr_out) {
            // fused LN finalize (GemmArgs::ln_mr_out): the last workgroup of this row block to get here reads the block's partials (all column tiles')
            // and writes (mean, rstd) - 8 lanes per row, quads j, j + 8, ..., then the 1-2-4 butterfly: exactly ln_finalize_kernel (elementwise.hip)
            // (no __threadfence(): a device-scope fence is an L2 writeback + invalidate per wave here.  The partials are system-scope write-through stores,
            //  acknowledged at the barrier's vmcnt(0); the counter and the reads below are system-scope atomics as well.)
            __syncthreads();
            int* flag = reinterpret_cast<int*>(smem);
            if (tid == 0) {
                int* c = g.ln_cnt + bm;
                const int last = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == nbn - 1;
                if (last) __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                *flag = last;
            }
            __syncthreads();
            if (*flag) {
                const int NP = g.N >> 5;
                const float invD = 1.f / (float)g.N;
                const int j8 = tid & 7;
                for (int r = tid >> 3; r < BM; r += NT / 8) {
                    const long row = (long)m0 + r;
                    float s1 = 0.f, s2 = 0.f;
                    if (row < g.M) {
                        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(g.ln_part + row * (long)NP * 2);
                        for (int qd = j8; qd < NP / 2; qd += 8) {
                            const f32x2 t0 = __builtin_bit_cast(f32x2, __hip_atomic_load(p + 2 * qd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                            const f32x2 t1 = __builtin_bit_cast(f32x2, __hip_atomic_load(p + 2 * qd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                            s1 += t0[0]; s2 += t0[1];
                            s1 += t1[0]; s2 += t1[1];
                        }
                    }
#pragma unroll
                    for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
                    if (row < g.M && j8 == 0) {
                        const float mean = s1 * invD;
                        const float var = fmaxf(fmaf(-mean, mean, s2 * invD), 0.f);
                        *reinterpret_cast<f32x2*>(g.ln_mr_out + 2 * row) = f32x2{mean, rsqrtf(var + 1e-6f)};
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = m0 + (wm * TM + i) * 32 + l31;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nb = n0 + (wn * TN + j) * 32 + 4 * hi;
            epilogue_tile32<T>(g, m, nb, hi, acc[i][j]);
        }
    }
}

template <typename T, int WM, int WN, int TM, int TN, int NSTAGE>
static int launch_glds(const GemmArgs& g, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int smem = NSTAGE * (BM + BN) * 128;
    constexpr auto kern = gemm_glds_kernel<T, WM, WN, TM, TN, NSTAGE>;
    if (int rc = set_dyn_lds<kern>(smem)) return rc;
    const long nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(64 * WM * WN), smem, st, g);
    return (int)hipGetLastError();
}

template <typename T, int WM, int WN, int TM, int TN, int AMODE>
static int launch_cfg(const GemmArgs& g, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int smem = 2 * (BM + BN) * 128;
    constexpr auto kern = gemm_kernel<T, WM, WN, TM, TN, AMODE>;
    if (int rc = set_dyn_lds<kern>(smem)) return rc;
    const long nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(64 * WM * WN), smem, st, g);
    return (int)hipGetLastError();
}

// ---- runtime tuning switches: environment (MOGE_<KEY>) overridden by moge_tune_set(key, value) -------------------------
// Process-global and shared by every handle: guarded by a mutex (two handles on two host threads = one GPU each is a supported
// deployment).  A switch that the ENVIRONMENT moves off its default is announced once on stderr: production must not be re-routed
// to a non-default kernel silently.
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <cstdio>
static std::map<std::string, int>& tune_table() { static std::map<std::string, int> t; return t; }
static std::mutex& tune_mutex() { static std::mutex m; return m; }
int moge_tune_get(const char* key, int dflt) {
    std::lock_guard<std::mutex> lk(tune_mutex());
    auto& t = tune_table();
    auto it = t.find(key);
    if (it != t.end()) return it->second;
    const std::string env = std::string("MOGE_") + key;
    const char* e = getenv(env.c_str());
    const int v = e ? atoi(e) : dflt;
    if (e && v != dflt) fprintf(stderr, "[libmoge_hip] %s=%d overrides the tuned default (%d): non-default kernel selection\n", env.c_str(), v, dflt);
    t[key] = v;                        // cache: launches look switches up on every call
    return v;
}
extern "C" void moge_tune_set(const char* key, int value) {
    std::lock_guard<std::mutex> lk(tune_mutex());
    tune_table()[key] = value;
}

template <typename T, int AMODE>
static int launch_by_n(const GemmArgs& g, hipStream_t st) {
    const int g_conv_bm256 = moge_tune_get("CONV_BM256", 0);
    if (g.N > 64) return launch_cfg<T, 2, 2, 2, 2, AMODE>(g, st);     // 128 x 128
    if (g_conv_bm256 && g.M >= 4096) {
        if (g.N > 32) return launch_cfg<T, 4, 1, 2, 2, AMODE>(g, st); // 256 x 64
        return launch_cfg<T, 4, 1, 2, 1, AMODE>(g, st);               // 256 x 32
    }
    if (g.N > 32) return launch_cfg<T, 4, 1, 1, 2, AMODE>(g, st);     // 128 x 64
    return launch_cfg<T, 4, 1, 1, 1, AMODE>(g, st);                   // 128 x 32
}

// latency regime (batch 1): fewer than half a wave of 256x256 tiles leaves most CUs idle; the 128x128 / 64x128 kernels below have 4-8x the
// workgroups (measured at M = 3601: proj 24 vs 33 us, fc2 56 vs 90 us)
bool gemm_runs_pp(const GemmArgs& g) {
    if (!moge_tune_get("GEMM_PP", 1) || !gemm_pp_eligible(g)) return false;
    const long tiles = ((long)(g.M + 255) / 256) * ((g.N + 255) / 256);
    // crossover (M = 3601 per image, N = 1024): the 64 x 128 kernel takes 47 / 18 us per image (fc2 / proj), a 256 x 256 tile 74 / 28 us however few of
    // them there are - from ~94 tiles (two images: 116) the throughput kernel wins (batch 2: 169.3 -> 173.4 img/s)
    return tiles >= moge_tune_get("PP_MIN_TILES", 96);
}

// LN-fold producers (EPI_RESID + ln_part) that go to gemm_glds_kernel can finalise the statistics themselves (GemmArgs::ln_mr_out): true when launch_gemm<f16>
// would take that kernel for g - the caller then sets ln_mr_out / ln_cnt and skips launch_ln_finalize.
bool gemm_fuses_ln_finalize(const GemmArgs& g) {
    if (!moge_tune_get("LN_FINALIZE_FUSED", 1)) return false;
    if (g.epi != EPI_RESID || !g.ln_part || !g.x16 || (g.N & 63) != 0) return false;
    if (gemm_runs_pp(g)) return false;
    return g.N > 64 && !g.relu_in && (g.K % (8 * TT<f16>::CH)) == 0 && !moge_tune_get("DISABLE_GLDS", 0);
}

template <typename T>
int launch_gemm(const GemmArgs& g, int amode, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return -1;
    const int g_disable_glds = moge_tune_get("DISABLE_GLDS", 0);
    if ((g.K % TT<T>::CH) != 0 || (g.N % 4) != 0) return -1;
    if (amode != AMODE_LINEAR && (g.C % TT<T>::CH) != 0) return -1;
    switch (amode) {
    case AMODE_LINEAR:
        if constexpr (std::is_same<T, f16>::value) {
            if (gemm_runs_pp(g)) return launch_gemm_pp(g, st);
        }
        if (g.N > 64 && !g.relu_in && (g.K % (8 * TT<T>::CH)) == 0 && !g_disable_glds) {
            // latency regime: when 128x128 tiles do not even give every CU two workgroups, halve the tile rows (64x128; a 3-slab
            // ring = 72 KiB LDS, two co-resident workgroups per CU: at batch 1-2 the weights arrive cold from HBM and the second slab in flight is worth
            // more than a third workgroup - 7.3 -> 6.85 ms per image against the double buffer).  Same MFMA and K order: results are bit-identical.
            const long blocks128 = ((long)(g.M + 127) / 128) * ((g.N + 127) / 128);
            if (blocks128 < moge_tune_get("GLDS_SMALL_BLOCKS", 512) && moge_tune_get("GLDS_VARIANT", 2) == 2) {
                switch (moge_tune_get("GLDS_SMALL_NS", 3)) {            // ring depth of the 64x128 kernel (bit-identical results)
                case 3: return launch_glds<T, 2, 2, 1, 2, 3>(g, st);
                case 4: return launch_glds<T, 2, 2, 1, 2, 4>(g, st);
                default: return launch_glds<T, 2, 2, 1, 2, 2>(g, st);
                }
            }
            switch (moge_tune_get("GLDS_VARIANT", 2)) {
            case 1: return launch_glds<T, 2, 2, 2, 2, 1>(g, st);       // 128x128, single buffer
            case 2: return launch_glds<T, 2, 2, 2, 2, 2>(g, st);       // 128x128, double buffer
            case 4: return launch_glds<T, 2, 2, 2, 2, 3>(g, st);       // 128x128, 3-slab ring
            case 5: return launch_glds<T, 2, 2, 2, 2, 4>(g, st);       // 128x128, 4-slab ring
            default: return launch_glds<T, 4, 2, 2, 2, 3>(g, st);      // 256x128, 8 waves, 3-slab ring
            }
        }
        return launch_by_n<T, AMODE_LINEAR>(g, st);
    case AMODE_CONV3:
        if constexpr (std::is_same<T, f16>::value) {
            if (moge_tune_get("CONV_PP", 1) && conv_pp_eligible(g)) return launch_conv_pp(g, st);
        }
        // the fused output conv (GemmArgs::dot_tab, g.out == nullptr) exists in the halo kernel only: any other kernel would store the full
        // map through a null pointer.  Unsupported shape, not a launch.
        if (g.dot_tab) return -1;
        return launch_by_n<T, AMODE_CONV3>(g, st);
    }
    return -1;
}

template int launch_gemm<f16>(const GemmArgs&, int, hipStream_t);
template int launch_gemm<float>(const GemmArgs&, int, hipStream_t);
