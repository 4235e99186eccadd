// Shared device/host definitions for libmoge_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef _Float16 f16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// --------------------------------------------------------------------------------------------
// Storage-type traits.  Everything matrix-shaped moves in 16-byte "chunks" along the contraction
// axis: 8 halves or 4 floats.  One "k-step" of the 32x32 MFMA consumes two chunks (lane half hi = lane>>5
// takes chunk 2*s+hi), which is ONE v_mfma_f32_32x32x16_f16 or FOUR v_mfma_f32_32x32x2_f32.
// C/D layout of both: col j = lane&31, row i = (r&3) + 8*(r>>2) + 4*(lane>>5), r in [0,16).
// --------------------------------------------------------------------------------------------
template <typename T> struct TT;
template <> struct TT<f16> {
    static constexpr int CH = 8;            // elements per 16-byte chunk
    static constexpr int PREC = 1;
};
template <> struct TT<float> {
    static constexpr int CH = 4;
    static constexpr int PREC = 0;
};

template <typename T>
__device__ __forceinline__ void mma_step(f32x16& acc, const u32x4& a, const u32x4& b);
template <>
__device__ __forceinline__ void mma_step<f16>(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_step<float>(f32x16& acc, const u32x4& a, const u32x4& b) {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
}

// v_mfma_f32_16x16x32_f16: A rows / B columns = lane & 15, K group (8 halves) = lane >> 4; D: column = lane & 15, rows 4*(lane >> 4) + r.
// Measured on this chip (tools/mfma_power.cpp, random operands): bare chains of this instruction sustain 1.90 PFLOP/s at 1.94 GHz,
// the 32x32x16 form 1.43-1.62 at 1.66 GHz - the chip clocks to its power budget and the 16x16 form is the cheaper one per FLOP.
template <typename T>
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b);
template <>
__device__ __forceinline__ void mma16<f16>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

// cross-lane moves on the VALU (DPP): lane ^ 1 and lane ^ 2 inside a quad, lane -> 7 - lane inside 8 lanes (all lanes active)
__device__ __forceinline__ float dpp_quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_half_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }

// the first MFMA of an accumulation chain: C = the inline constant 0 (no zero fill of the accumulator registers)
template <typename T>
__device__ __forceinline__ f32x4 mma16_first(const u32x4& a, const u32x4& b);
template <>
__device__ __forceinline__ f32x4 mma16_first<f16>(const u32x4& a, const u32x4& b) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}

// row of the 32x32 accumulator tile held in register r by a lane of half `hi`
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// convert 4 floats to storage type and store (8 B for f16, 16 B for f32); dst 8/16-byte aligned
__device__ __forceinline__ void store4(f16* dst, float a, float b, float c, float d) {
    f16x4 v = {(f16)a, (f16)b, (f16)c, (f16)d};
    *reinterpret_cast<f16x4*>(dst) = v;
}
__device__ __forceinline__ void store4(float* dst, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    *reinterpret_cast<f32x4*>(dst) = v;
}
__device__ __forceinline__ void load4(const f16* src, float* o) {
    f16x4 v = *reinterpret_cast<const f16x4*>(src);
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
}
__device__ __forceinline__ void load4(const float* src, float* o) {
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}

// elementwise ReLU on a 16-byte chunk of storage type T
template <typename T> __device__ __forceinline__ u32x4 relu_chunk(u32x4 v);
template <> __device__ __forceinline__ u32x4 relu_chunk<f16>(u32x4 v) {
    f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = h[i] > (f16)0 ? h[i] : (f16)0;
    return __builtin_bit_cast(u32x4, h);
}
template <> __device__ __forceinline__ u32x4 relu_chunk<float>(u32x4 v) {
    f32x4 h = __builtin_bit_cast(f32x4, v);
#pragma unroll
    for (int i = 0; i < 4; i++) h[i] = fmaxf(h[i], 0.f);
    return __builtin_bit_cast(u32x4, h);
}

// LDS swizzle of the chunk index inside a tile row.  CPR = chunks per row (8 -> 128-byte rows, 16 -> 256-byte rows).
// ds_read_b128 is served in 16-lane groups over a 256-byte bank row: with these XORs the 16 lanes of a group
// (16 different rows, same logical chunk) hit 16 distinct 16-byte slots.
template <int CPR> __device__ __forceinline__ int swz(int row, int c);
template <> __device__ __forceinline__ int swz<8>(int row, int c) { return c ^ ((row >> 1) & 7); }
template <> __device__ __forceinline__ int swz<16>(int row, int c) { return c ^ (row & 15); }

// force a wave-uniform pointer onto the scalar unit (defeats per-lane 64-bit pointer induction variables in DMA loops)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// gelu(x) = x * Phi(x) with Phi(x) = sigmoid(x * P(x^2)); P fitted (weighted minimax, degree 4 in x^2) to
// logit(Phi(x))/x on |x| <= 5, max abs error of gelu 3.7e-6 over all x (P grows for |x| > 5, so both tails saturate
// correctly: exp2 -> 0 or +inf).  8 VALU + 2 transcendentals per element instead of erff's ~35.  fp16 outputs only.
__device__ __forceinline__ float gelu_fast(float x) {
    constexpr float L2E = 1.4426950408889634f;
    const float x2 = x * x;
    float p = 2.09755530e-06f * L2E;
    p = fmaf(p, x2, -5.83663339e-05f * L2E);
    p = fmaf(p, x2, -2.66659641e-04f * L2E);
    p = fmaf(p, x2, 7.29729188e-02f * L2E);
    p = fmaf(p, x2, 1.59563637f * L2E);
    const float e = __builtin_amdgcn_exp2f(-(p * x));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// two elements at a time: the polynomial runs on packed-fp32 VALU ops (v_pk_mul_f32 / v_pk_fma_f32), the two transcendentals
// stay scalar.  Same operation sequence per element as gelu_fast (bit-identical results).
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    constexpr float L2E = 1.4426950408889634f;
    const f32x2 x2 = x * x;
    f32x2 p = {2.09755530e-06f * L2E, 2.09755530e-06f * L2E};
    p = __builtin_elementwise_fma(p, x2, f32x2{-5.83663339e-05f * L2E, -5.83663339e-05f * L2E});
    p = __builtin_elementwise_fma(p, x2, f32x2{-2.66659641e-04f * L2E, -2.66659641e-04f * L2E});
    p = __builtin_elementwise_fma(p, x2, f32x2{7.29729188e-02f * L2E, 7.29729188e-02f * L2E});
    p = __builtin_elementwise_fma(p, x2, f32x2{1.59563637f * L2E, 1.59563637f * L2E});
    const f32x2 t = -(p * x);
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    const f32x2 d = e + f32x2{1.0f, 1.0f};
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return x * r;
}
// Epilogue arithmetic shared by gemm.hip and gemm_pp.hip, written with explicit fmaf so that both kernels round identically:
// the fp16 path picks one or the other by problem size (latency regime), and a batch item must not depend on its batch.
__device__ __forceinline__ float resid_term(float gamma, float acc, float bias) { return fmaf(gamma, acc, gamma * bias); }   // gamma*(acc+bias)
__device__ __forceinline__ float uv_term_add(float v, float wu, float u, float wv, float vv) { return fmaf(wu, u, fmaf(wv, vv, v)); }

// q pre-scale of the QKV epilogues: the product is made opaque so that no kernel fuses it with the following fp16 conversion into one
// v_fma_mixlo_f16 (single rounding) while another emits v_mul_f32 + v_cvt (double rounding): 3e-5 of the q entries differed by one fp16 ulp
// between gemm.hip and gemm_pp.hip before this
__device__ __forceinline__ float q_scaled(float v, float s) { float r = v * s; asm volatile("" : "+v"(r)); return r; }

// LN fold (GemmArgs::ln_mr): one output element of the consumer GEMM, and the (sum, sum of squares) of 4 consecutive residual columns
__device__ __forceinline__ float ln_fold_term(float acc, float mean, float rstd, float c, float b) { return fmaf(rstd, fmaf(-mean, c, acc), b); }
__device__ __forceinline__ void ln_quad_sums(const f32x4& x, float& s1, float& s2) {
    s1 = (x[0] + x[1]) + (x[2] + x[3]);
    s2 = fmaf(x[1], x[1], x[0] * x[0]) + fmaf(x[3], x[3], x[2] * x[2]);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// torch.linspace(start, end, steps) element i in fp32 (ATen: symmetric evaluation around the middle)
__device__ __forceinline__ float linspace_at(float start, float end, float step, int steps, int i) {
    return (i < steps / 2) ? (start + step * (float)i) : (end - step * (float)(steps - 1 - i));
}

// --------------------------------------------------------------------------------------------
// GEMM / implicit-GEMM conv argument blocks (see gemm.hip)
// --------------------------------------------------------------------------------------------
enum { AMODE_LINEAR = 0, AMODE_CONV3 = 1 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_PATCH = 2, EPI_QKV = 3, EPI_CONVT = 4 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

struct UVTerm {            // out[n] += wu[n]*u(x) + wv[n]*v(y): the 1x1 conv of the (u,v) planes (v2.py:154-160)
    const float* wu;       // null -> disabled
    const float* wv;
    float u0, u1, ustep, v0, v1, vstep;
};

struct GemmArgs {
    // A operand (activations)
    const void* a;
    int lda;               // LINEAR: row stride in elements
    int H, W, C;           // CONV3: spatial dims and input channels of the NHWC input
    int relu_in;           // apply ReLU to A on load
    const void* a2;        // conv_pp only: fused 1x1 side input (same NHWC shape as a), out += w2[N][C] . a2;  null -> none
    const void* w2;
    // W operand: [N][ldw] storage type, K-contiguous
    const void* w;
    int ldw;
    int M, N, K;           // K in elements (multiple of the chunk size)
    // epilogue
    int epi;
    int act;
    const float* bias;     // [N] fp32 or null
    void* out;             // storage type (EPI_STORE / EPI_CONVT)
    int ldc;
    const void* add;       // storage type, same indexing as out (EPI_STORE), or null
    int ldadd;
    UVTerm uv;
    int pixW, pixH;        // spatial dims of a row index m = (b*pixH + y)*pixW + x  (uv term, EPI_CONVT)
    // EPI_RESID: x[m*ldc+n] += gamma[n]*(acc+bias[n])   (fp32 residual stream)
    float* xres;
    const float* gamma;
    // EPI_PATCH: xres[(b*Ntok+1+p)*N + n] = acc + bias[n] + pos[(1+p)*N + n]; with cls: the rows of patch 0 also write the image's cls row xres[b*Ntok*N + n] = cls[n] + pos[n]
    const float* pos;
    const float* cls;
    int Np, Ntok;
    // EPI_QKV: scatter to q,k (B,nh,Ntok,64) and vT (B,nh,64,Npad); q pre-scaled by qscale
    void* q; void* k; void* vT;
    int v_rowmajor;        // 1: V goes to vT as (B,nh,Ntok,64) like K (attention_pp); 0: transposed (B,nh,64,Npad)
    int nh, Npad, D;
    float qscale;
    // EPI_CONVT: n = (dy*2+dx)*Cout + co -> out[((b*2H+2y+dy)*2W+2x+dx)*Cout+co]
    int Cout;
    // LayerNorm folded into the GEMMs around it (fp16 path; model.hip "LN fold"):  LN(x) W^T + b = rstd (x W'^T - mean c) + b'
    //   with W' = g (.) W (fp16),  c[n] = sum_k W'[n][k],  b' = b + W beta.
    //   producer (EPI_RESID): x16[m*ldc+n] = fp16 copy of the UPDATED residual, ln_part[(m*(N/32) + n/32)*2 + {0,1}] = (sum, sum of
    //     squares) of its 32-column group (fixed summation tree, identical in gemm.hip and gemm_pp.hip)
    //   consumer (EPI_QKV / EPI_STORE): A = x16, W = W', bias = b', ln_mr[2m + {0,1}] = (mean, rstd) of row m, ln_c = c
    void* x16;
    float* ln_part;
    const float* ln_mr;
    const float* ln_c;
    int uv_in;             // EPI_CONVT + uv: the term is evaluated at the INPUT pixel m with wu / wv indexed by the GEMM column n (MoGe-1: the uv
                           // channels are inputs of the ConvTranspose2d, v1.py:118-121); 0: at the output pixel, indexed by output channel (MoGe-2 resampler)
    int nt_store;          // gemm_pp: non-temporal stores for the f16 output rows (streaming activations; keeps the residual in the Infinity Cache)
    unsigned long long* dbg_ts;   // gemm_pp128: optional s_memtime stamps of one block (tools/kbench)
    int stagger;           // gemm_pp128p_kernel: > 1 = the workgroups of an XCD start in that many phase groups spread over one tile time (PP_STAGGER; 0 / 1 = off)
    int dbg;               // gemm_pp ablation bits (tools/kbench only): 1 no DMA, 2 no LDS reads, 4 no MFMA, 8 no barriers
    int stagger_clk;       // ... and over at most this many shader clocks (PP_STAGGER_CLK)
    int abl;               // gemm_pp128p_kernel, -DMOGE_EXPERIMENTS builds only (PP_ABL, tools/energy.sh): 1 no main-loop DMA, 2 fragment reads only in K-tile 0, 4 no epilogue, 8 no MFMA
    // conv_rb.hip (fused residual block, modules.py:47-68): out = add + conv2(relu(conv1(relu(a)) + bias)) + rb_bias2; w / rb_w2 = [C][9C]
    const void* rb_w2;
    const float* rb_bias2;
    // conv_pp.hip EPI_CONVT with a fused 1x1 output conv (level 4 of the decoder, fp16 path): the 32 channels of every high-res pixel are
    // rounded to fp16 and contracted with dot_nd groups of four output rows on the matrix pipe; the (B, 2H, 2W, Cout) map is NOT stored.
    //   dot_out[((b*2H + Y)*2W + X) * 4*dot_nd + 4*gq + e] = sum_c dot_tab(gq, e, c) * fp16(x4[b, Y, X, c])        (fp32)
    // dot_tab: f16 [dot_nd][64 lanes][8], the rows in MFMA A-operand lane layout (pack_dot_table in elementwise.hip)
    const void* dot_tab;
    float* dot_out;
    int dot_nd;
    // conv_pp.hip CT3 (round 6): ConvTranspose2d(k2, s2) + the 3x3 conv behind it as one 4-phase 2x2-tap conv on the low-res map: a = (B, H, W, C) low-res input,
    // w = composed weights [4 Cout][4 C] (row = phase * Cout + co, column = tap * C + ci), N = 4 Cout, K = 4 C, epi = EPI_CONVT, a2 = optional HIGH-res side map
    // (B, 2H, 2W, Cout) with w2 [Cout][Cout].  The border ring is corrected afterwards by launch_ct3_border.
    int ct3;
    // LN-fold producer, latency regime (gemm_glds_kernel only; round 6): the workgroup that completes a row block's last column tile turns the block's
    // (sum, sum of squares) partials into (mean, rstd) itself - ln_finalize_kernel's arithmetic in ln_finalize_kernel's order, bit-identical - so that the
    // separate 5 us launch between producer and consumer disappears at batch 1.  ln_cnt: one int per 64 rows, zero before the launch, left zero.
    float* ln_mr_out;
    int* ln_cnt;
};

// Opt a kernel into more than 64 KiB of dynamic LDS.  The attribute is per DEVICE and a process may drive several (one host thread per
// GPU is a supported deployment): remember it per (kernel, device), not per process.
template <auto Kern>
inline int set_dyn_lds(int bytes) {
    static bool done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64 || !done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
    return 0;
}

// CU count of the CURRENT device (persistent kernels size their grids with it).  Cached per device id: a process may drive several GPUs,
// and partition modes (SPX / CPX) give devices of one box different counts.
inline int pp_device_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64 && cus[dev]) return cus[dev];
    hipDeviceProp_t p;
    const int n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    if (dev >= 0 && dev < 64) cus[dev] = n;
    return n;
}

// host-side launchers (gemm.hip, gemm_pp.hip)
template <typename T> int launch_gemm(const GemmArgs& g, int amode, hipStream_t st);
bool gemm_pp_eligible(const GemmArgs& g);
void moge_internal_set_error(const char* msg);      // model.hip: text behind moge_last_error()
bool gemm_fuses_ln_finalize(const GemmArgs& g);      // launch_gemm<f16>(g, AMODE_LINEAR) will take gemm_glds_kernel, which can finalise the LN statistics itself (GemmArgs::ln_mr_out)
bool gemm_runs_pp(const GemmArgs& g);       // launch_gemm<f16>(g, AMODE_LINEAR) will take the ping-pong throughput kernel (profiler class)
int launch_gemm_pp(const GemmArgs& g, hipStream_t st);
bool conv_pp_eligible(const GemmArgs& g);
int launch_conv_pp(const GemmArgs& g, hipStream_t st);
// one composed weight block of the CT3 forms = a signed sum of basic products P(k, s) (elementwise.hip ct3_combine_kernel): term = k | s << 4 | (negative ? 256 : 0)
struct Ct3Slot { int rowblk, colblk, nterms; int term[13]; };
int launch_ct3_combine(const float* P, const Ct3Slot* slots_dev, int nslots, int Cout, int Cin, void* out, int ld, hipStream_t st);
// CT3 border: out (B, 2H, 2W, Cout) f16 += difference between replicate padding and what the composed conv computes on the outermost ring of output pixels;
// dw = border weights [12 classes][Cout][2 Cin] f16 (elementwise.hip compose_ct3), in = low-res input (B, H, W, Cin) f16
int launch_ct3_border(const void* in, const void* dw, void* out, int B, int H, int W, int Cin, int Cout, hipStream_t st);
bool conv_rb_eligible(const GemmArgs& g);   // conv_rb.hip: fp16 residual block with the intermediate map kept in LDS (C = 64)
int launch_conv_rb(const GemmArgs& g, hipStream_t st);
// runtime tuning / A-B switches (tests, tools/kbench): see tune.cpp-style table in gemm.hip
int moge_tune_get(const char* key, int dflt);
