// "Ping-pong" MFMA GEMM for the big fp16 linear layers of the MoGe-2 hot path (gfx950 / CDNA4).
//
//   C[M,N] = A[M,K] * W[N,K]^T      A: fp16 activations [M][lda], W: fp16 weights [N][ldw], both K-contiguous
//
// Serves (f16 mode): ViT qkv / proj / fc1 / fc2 (attention.py:72,79  mlp.py:35,38), the summed output projections
// (modules.py:128-131), the level-0 1x1 input blocks and the ConvTranspose2d-as-GEMM resamplers (modules.py:162,245).
// The fp32 parity mode and every shape this kernel does not take stay on gemm.hip.
//
// Structure (one workgroup per CU, 8 waves = 2 per SIMD, 256 x BN output tile, BN = 256 or 128):
//   * K is consumed in PHASES of 32 halves (64 bytes per tile row).  A phase's A and W rows live in one LDS slot
//     [BM+BN rows][64 B]; four slots form a ring, filled by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip).
//     One wave-instruction fills 16 rows (1 KiB, lane-linear in LDS); the XOR swizzle that makes ds_read_b128
//     conflict-free (chunk ^ ((row>>2)&3)) is applied to the per-lane SOURCE chunk and to the read address.
//   * The two waves of every SIMD are in different wave GROUPS (waves 0-3 / 4-7) that run half a phase apart:
//     while one group issues its 16 MFMAs (32x32x16, "compute segment"), the other reads its fragments from LDS and
//     issues the DMA for phase+3 ("load segment"); an s_barrier separates the segments, so each SIMD's matrix pipe
//     always has exactly one wave feeding it and LDS / DMA latency sits under the partner's MFMAs.
//   * DMA completion is tracked with COUNTED s_waitcnt vmcnt(N) (two phases stay in flight across the barriers);
//     raw s_barrier only - __syncthreads() would drain the DMA queue.
//   * The MFMA is issued "swapped" (A-operand = weight rows) so a lane owns ONE output row and 4 consecutive
//     columns per register quad.  The epilogue applies bias / activation / LayerScale in registers, transposes the
//     wave's 128x64 (or 64x64) sub-tile through its private LDS region and writes / read-modify-writes global memory
//     in full 128-byte row segments (row-per-lane stores cost one cache-line lookup per lane on this chip).
#include "common.h"
#include <type_traits>

#define PP_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
// A per-lane 32-bit DMA source offset, made opaque AT THE POINT OF USE: the add  uniform base + zext(offset)  then stays inside the loop and selects the
// SADDR form  global_load_lds_dwordx4 vOff, s[base:base+1]  (no VALU).  Left to itself the compiler hoists the zero-extension out of the K loop, keeps
// every offset as a 64-bit VGPR pair and pays one v_lshl_add_u64 per piece and tile in the read segments (round 5: 8 VALU + 8 VGPRs per wave and K-tile).
__device__ __forceinline__ unsigned pp_opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }
#define PP_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- shared epilogue: this wave's (TM*32) x 64 accumulator tile -> global memory ---------------------------------
// Every wave has passed the final barrier: all LDS reads and all DMA writes of the ring are complete, so each wave may
// reuse its private 16 KiB (TM = 4) / 8 KiB (TM = 2) region of the ring for the output transpose.
// The flavour is a COMPILE-TIME parameter: with run-time switches inside the 128-accumulator loops the compiler emitted
// ~1000 register copies and ~100 branches per wave (measured 13-45k cycles per tile against a 43k-cycle K=1024 main loop).
enum { EPK_RESID = 0, EPK_STORE = 1, EPK_GELU = 2, EPK_QKV = 3, EPK_CONVT = 4, EPK_UV = 5, EPK_RELU = 6,
       EPK_GELU_LN = 7, EPK_QKV_LN = 8,          // _LN: consumer of a folded LayerNorm (GemmArgs::ln_mr), otherwise as GELU / QKV
       EPK_RESID16 = 9 };                        // RESID on an fp16 residual stream (GemmArgs::xres == nullptr: x16 IS the stream; `.half()` models)
template <int EPK> struct EpkBase { static constexpr int K = EPK == EPK_GELU_LN ? EPK_GELU : (EPK == EPK_QKV_LN ? EPK_QKV : EPK);
                                    static constexpr bool FOLD = EPK == EPK_GELU_LN || EPK == EPK_QKV_LN; };

// ---- stage 2: LDS staging region (row-major, 128 B per row, 16-byte chunks XOR-swizzled by row & 7) -> global memory ----
// The read-modify-write of the fp32 residual is split into three straight-line stages so that NO load is ever waited for behind a store:
// on gfx950 vmcnt counts stores as well as loads and retires in issue order, so a loop of {wait for row i's load; add; store row i}
// makes every iteration wait for the previous iteration's STORES to be acknowledged (measured: ~1 us per row group, 18 us of a 43 us
// K = 1024 tile).  Here: (1) all row loads of a pass are issued, (2) all adds are done (waits see loads only), (3) all stores are issued
// back to back; the caller issues the NEXT pass's loads before this pass's stores.
// Addressing: raw buffer instructions over a descriptor of exactly M rows (wave-uniform, built from kernel arguments) + ONE 32-bit byte
// offset per lane - no 64-bit per-lane address lives across the stages (with them the kernel spilled, and every scratch reload waited
// vmcnt(0) = for all stores in flight); rows >= M are out of range of the descriptor: their loads return 0 and their stores are dropped.
struct ResidBufs {
    __amdgpu_buffer_rsrc_t x, x16, part;
    unsigned ldc4, ldc2, npart8;        // row pitches in bytes: fp32 residual, fp16 copy, (sum, sum of squares) pairs
};
__device__ __forceinline__ ResidBufs pp_resid_bufs(const GemmArgs& g) {
    ResidBufs r;
    r.ldc4 = (unsigned)g.ldc * 4u; r.ldc2 = (unsigned)g.ldc * 2u; r.npart8 = (unsigned)(g.N >> 5) * 8u;
    r.x = __builtin_amdgcn_make_buffer_rsrc(g.xres, 0, (int)((unsigned)g.M * r.ldc4), 0x00020000);
    r.x16 = __builtin_amdgcn_make_buffer_rsrc(g.x16, 0, g.x16 ? (int)((unsigned)g.M * r.ldc2) : 0, 0x00020000);
    r.part = __builtin_amdgcn_make_buffer_rsrc(g.ln_part, 0, g.ln_part ? (int)((unsigned)g.M * r.npart8) : 0, 0x00020000);
    return r;
}
template <int WROWS>
__device__ __forceinline__ void pp_resid_load(const ResidBufs& rb, int lane, int mw, int ncol, f32x4 (&xv)[WROWS / 8]) {
    const int rr = lane >> 3, cc = lane & 7;
    const unsigned off0 = (unsigned)(mw + rr) * rb.ldc4 + (unsigned)(ncol + cc * 4) * 4u;
#pragma unroll
    for (int it = 0; it < WROWS / 8; it++)
        xv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb.x, (int)(off0 + (unsigned)it * 8u * rb.ldc4), 0, 0));
}
template <int WROWS>
__device__ __forceinline__ void pp_resid_add(const char* R, int lane, f32x4 (&xv)[WROWS / 8]) {
    const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
    for (int it = 0; it < WROWS / 8; it++) {
        const int row = it * 8 + rr;
        const f32x4 v = *reinterpret_cast<const f32x4*>(R + row * 128 + ((cc ^ (row & 7)) << 4));
        xv[it] = xv[it] + v;
    }
}
template <int WROWS>
__device__ __forceinline__ void pp_resid_store(const ResidBufs& rb, bool fold, int lane, int mw, int ncol, const f32x4 (&xv)[WROWS / 8]) {
    const int rr = lane >> 3, cc = lane & 7;
    const unsigned off0 = (unsigned)(mw + rr) * rb.ldc4 + (unsigned)(ncol + cc * 4) * 4u;
#pragma unroll
    for (int it = 0; it < WROWS / 8; it++)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xv[it]), rb.x, (int)(off0 + (unsigned)it * 8u * rb.ldc4), 0, 0);
    if (fold) {               // LN fold producer: fp16 copy + (sum, sum of squares) of each row's 32-column group (8 lanes, butterfly 1-2-4)
        // lane pairs (cc, cc ^ 1) share one 16-byte store of 8 columns: odd lanes point past the end of the descriptor (store dropped)
        const unsigned h0 = (cc & 1) ? 0xffffff00u : (unsigned)(mw + rr) * rb.ldc2 + (unsigned)(ncol + cc * 4) * 2u;
        const unsigned p0 = cc ? 0xffffff00u : (unsigned)(mw + rr) * rb.npart8 + (unsigned)(ncol >> 5) * 8u;
#pragma unroll
        for (int it = 0; it < WROWS / 8; it++) {
            const f32x4 xnew = xv[it];
            const f16x4 h4 = {(f16)xnew[0], (f16)xnew[1], (f16)xnew[2], (f16)xnew[3]};
            const u32x2 mine = __builtin_bit_cast(u32x2, h4);
            u32x4 pk;
            pk[0] = mine[0]; pk[1] = mine[1];
            pk[2] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine[0], 0xB1, 0xF, 0xF, true);          // lane ^ 1's halves (DPP quad_perm [1,0,3,2])
            pk[3] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine[1], 0xB1, 0xF, 0xF, true);
            __builtin_amdgcn_raw_buffer_store_b128(pk, rb.x16, (int)((cc & 1) ? h0 : h0 + (unsigned)it * 8u * rb.ldc2), 0, 0);
            float s1, s2;
            ln_quad_sums(xnew, s1, s2);
            s1 += dpp_quad_xor1(s1); s2 += dpp_quad_xor1(s2);      // __shfl_xor(., 1 / 2 / 4) on the VALU: the quads are uniform before the last step,
            s1 += dpp_quad_xor2(s1); s2 += dpp_quad_xor2(s2);      // where lane i takes lane 7 - i's value (the other quad's sum)
            s1 += dpp_half_mirror(s1); s2 += dpp_half_mirror(s2);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{s1, s2}), rb.part, (int)(cc ? p0 : p0 + (unsigned)it * 8u * rb.npart8), 0, 0);
        }
    }
}

// ---- fp16 residual stream (EPK_RESID16) ----------------------------------------------------------------------------------------------
// `model.half()` keeps the residual stream itself in fp16 (block.py:110-112 on half tensors; scripts/infer.py:83-84): x16 is read, updated and
// written in place - 4 bytes per element instead of the 10 of the fp32 stream + fp16 copy - and the LN-fold statistics are taken from the
// ROUNDED values (the operand the consumer GEMM reads).  x_new = fp16(fp32(x_old) + gamma (acc + bias)): one rounding, where the reference
// rounds the projection, the LayerScale product and the sum separately.
// Staging: PROWS rows x 256 B (the wave's 64 fp32 columns), 16-byte chunks XOR-swizzled by row & 15; a lane owns 8 consecutive columns of
// rows rr, rr + 8, ...: one 16-byte load and one 16-byte store per row, full 128-byte row segments.  Summation tree of the statistics = the
// fp32 flavour's (quads, pairs of quads, 1-2-4 butterfly: here the first pair is lane-local), so gemm.hip's latency kernels agree bit for bit.
struct Resid16Bufs {
    __amdgpu_buffer_rsrc_t x16, part;
    unsigned ldc2, npart8;
};
__device__ __forceinline__ Resid16Bufs pp_resid16_bufs(const GemmArgs& g) {
    Resid16Bufs r;
    r.ldc2 = (unsigned)g.ldc * 2u; r.npart8 = (unsigned)(g.N >> 5) * 8u;
    r.x16 = __builtin_amdgcn_make_buffer_rsrc(g.x16, 0, (int)((unsigned)g.M * r.ldc2), 0x00020000);
    r.part = __builtin_amdgcn_make_buffer_rsrc(g.ln_part, 0, g.ln_part ? (int)((unsigned)g.M * r.npart8) : 0, 0x00020000);
    return r;
}
template <int PROWS>
__device__ __forceinline__ void pp_resid16_load(const Resid16Bufs& rb, int lane, int mw, int ncol, u32x4 (&xv)[PROWS / 8]) {
    const int rr = lane >> 3, cc = lane & 7;
    const unsigned off0 = (unsigned)(mw + rr) * rb.ldc2 + (unsigned)(ncol + cc * 8) * 2u;
#pragma unroll
    for (int it = 0; it < PROWS / 8; it++) xv[it] = __builtin_amdgcn_raw_buffer_load_b128(rb.x16, (int)(off0 + (unsigned)it * 8u * rb.ldc2), 0, 0);
}
// x <- fp16(x + staged update); returns the packed result in xv
template <int PROWS>
__device__ __forceinline__ void pp_resid16_add(const char* R, int lane, u32x4 (&xv)[PROWS / 8]) {
    const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
    for (int it = 0; it < PROWS / 8; it++) {
        const int row = it * 8 + rr;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(R + row * 256 + (((2 * cc) ^ (row & 15)) << 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(R + row * 256 + (((2 * cc + 1) ^ (row & 15)) << 4));
        const f16x8 h = __builtin_bit_cast(f16x8, xv[it]);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 4; e++) { o[e] = (f16)((float)h[e] + v0[e]); o[4 + e] = (f16)((float)h[4 + e] + v1[e]); }
        xv[it] = __builtin_bit_cast(u32x4, o);
    }
}
template <int PROWS>
__device__ __forceinline__ void pp_resid16_store(const Resid16Bufs& rb, bool fold, int lane, int mw, int ncol, const u32x4 (&xv)[PROWS / 8]) {
    const int rr = lane >> 3, cc = lane & 7;
    const unsigned off0 = (unsigned)(mw + rr) * rb.ldc2 + (unsigned)(ncol + cc * 8) * 2u;
#pragma unroll
    for (int it = 0; it < PROWS / 8; it++) __builtin_amdgcn_raw_buffer_store_b128(xv[it], rb.x16, (int)(off0 + (unsigned)it * 8u * rb.ldc2), 0, 0);
    if (fold) {
        // one (sum, sum of squares) pair per row and 32-column group: lanes cc = 0 and 4 store, the others point past the descriptor
        const unsigned p0 = (cc & 3) ? 0xffffff00u : (unsigned)(mw + rr) * rb.npart8 + (unsigned)((ncol + cc * 8) >> 5) * 8u;
#pragma unroll
        for (int it = 0; it < PROWS / 8; it++) {
            const f16x8 h = __builtin_bit_cast(f16x8, xv[it]);
            const f32x4 a = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]}, b = {(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
            float a1, a2, b1, b2;
            ln_quad_sums(a, a1, a2);
            ln_quad_sums(b, b1, b2);
            float s1 = a1 + b1, s2 = a2 + b2;
            s1 += dpp_quad_xor1(s1); s2 += dpp_quad_xor1(s2);      // (what __shfl_xor(., 1 / 2) computes, on the VALU instead of ds_bpermute + wait)
            s1 += dpp_quad_xor2(s1); s2 += dpp_quad_xor2(s2);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{s1, s2}), rb.part, (int)((cc & 3) ? p0 : p0 + (unsigned)it * 8u * rb.npart8), 0, 0);
        }
    }
}
// stage the 16-row blocks i0 .. i0 + PROWS / 16 - 1 of a wave's accumulators (all 64 columns) into R
template <int PROWS, int NJT>
__device__ __forceinline__ void pp_resid16_stage(char* R, const f32x4 (&acc)[8][NJT], int j0, int i0, int lane) {
    const int l15 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int ii = 0; ii < PROWS / 16; ii++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int row = ii * 16 + l15;
            *reinterpret_cast<f32x4*>(R + row * 256 + (((jj * 4 + g4) ^ (row & 15)) << 4)) = acc[i0 + ii][j0 + jj];
        }
}

template <int WROWS, int EPK>
__device__ __forceinline__ void pp_store_rows(const GemmArgs& g, const char* R, int lane, int mw, int nw) {
    // 64 f16 columns per row, 16-byte stores; row base pointers of the three store flavours advance by rows of 8 per iteration
    const int rr = lane >> 3, cc = lane & 7;
    const int M = g.M;
    const int mfirst = mw + rr;
    if constexpr (EPK == EPK_QKV) {
        const int which = nw / g.D;
        const int head = (nw - which * g.D) >> 6;
        const int Ntok = g.Ntok;
        f16* obase = reinterpret_cast<f16*>(which == 0 ? g.q : (which == 1 ? g.k : g.vT)) + (size_t)head * Ntok * 64 + cc * 8;
        const size_t bstride = (size_t)g.nh * Ntok * 64;
        int b0 = mfirst / Ntok;
        int t0 = mfirst - b0 * Ntok;
#pragma unroll
        for (int it = 0; it < WROWS / 8; it++) {
            const int row = it * 8 + rr;
            const u32x4 v = *reinterpret_cast<const u32x4*>(R + row * 128 + ((cc ^ (row & 7)) << 4));
            f16* p = obase + (size_t)b0 * bstride + (size_t)t0 * 64;
            t0 += 8;
            if (t0 >= Ntok) { t0 -= Ntok; b0 += 1; }
            if (mw + row < M) { if (g.nt_store) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); else *reinterpret_cast<u32x4*>(p) = v; }
        }
    } else if constexpr (EPK == EPK_CONVT) {
        const int Cout = g.Cout, pixW = g.pixW, pixH = g.pixH;
        const int qd = nw / Cout;
        const int co0 = nw - qd * Cout, dy = qd >> 1, dx = qd & 1;
        f16* obase = reinterpret_cast<f16*>(g.out) + co0 + cc * 8;
        int px = mfirst % pixW;
        const int t = mfirst / pixW;
        int py = t % pixH, pb = t / pixH;
#pragma unroll
        for (int it = 0; it < WROWS / 8; it++) {
            const int row = it * 8 + rr;
            const u32x4 v = *reinterpret_cast<const u32x4*>(R + row * 128 + ((cc ^ (row & 7)) << 4));
            f16* p = obase + ((((size_t)pb * 2 * pixH + 2 * py + dy) * (2 * pixW)) + 2 * px + dx) * Cout;
            px += 8;
            if (px >= pixW) { px -= pixW; py += 1; if (py >= pixH) { py = 0; pb += 1; } }
            if (mw + row < M) *reinterpret_cast<u32x4*>(p) = v;
        }
    } else {
        f16* obase = reinterpret_cast<f16*>(g.out) + nw + cc * 8;
        const long ldc = g.ldc;
#pragma unroll
        for (int it = 0; it < WROWS / 8; it++) {
            const int row = it * 8 + rr;
            const int m = mw + row;
            const u32x4 v = *reinterpret_cast<const u32x4*>(R + row * 128 + ((cc ^ (row & 7)) << 4));
            if (m < M) { if (g.nt_store) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(obase + (size_t)m * ldc)); else *reinterpret_cast<u32x4*>(obase + (size_t)m * ldc) = v; }
        }
    }
}

// the per-quad arithmetic shared by both accumulator layouts: 4 consecutive columns n .. n+3 of one output row
// Written on PAIRS of columns: v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32 and one v_cvt_pk_f16_f32 per pair (the scalar form compiled to one VALU
// instruction per element and, in the plain-store flavour, to v_cvt_f16_f32 x 2 + v_pack + v_alignbit per pair: the epilogue runs with the matrix pipe
// idle, 2 waves per SIMD, 4 clocks per instruction).  Per element the operations and their order are those of gemm.hip's scalar epilogues
// (bit-identical: packed fp32 instructions round like their scalar forms).
template <int EPK, bool FOLD = false>
__device__ __forceinline__ f16x4 pp_quad_f16(const f32x4& a, const f32x4& b, float scale, const f32x4& wu, float u, const f32x4& wv, float vv,
                                             float mean = 0.f, float rstd = 1.f, const f32x4& lc = f32x4{0.f, 0.f, 0.f, 0.f}) {
    f32x2 v[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const f32x2 ah = {a[2 * h], a[2 * h + 1]}, bh = {b[2 * h], b[2 * h + 1]};
        if constexpr (FOLD) {
            const f32x2 ch = {lc[2 * h], lc[2 * h + 1]};
            v[h] = __builtin_elementwise_fma(f32x2{rstd, rstd}, __builtin_elementwise_fma(f32x2{-mean, -mean}, ch, ah), bh);      // ln_fold_term
        } else {
            v[h] = ah + bh;
        }
        if constexpr (EPK == EPK_QKV) { v[h] = v[h] * f32x2{scale, scale}; asm volatile("" : "+v"(v[h])); }                        // q_scaled
        if constexpr (EPK == EPK_UV)
            v[h] = __builtin_elementwise_fma(f32x2{wu[2 * h], wu[2 * h + 1]}, f32x2{u, u},
                                             __builtin_elementwise_fma(f32x2{wv[2 * h], wv[2 * h + 1]}, f32x2{vv, vv}, v[h]));    // uv_term_add
        if constexpr (EPK == EPK_RELU) v[h] = f32x2{fmaxf(v[h][0], 0.f), fmaxf(v[h][1], 0.f)};
        if constexpr (EPK == EPK_GELU) v[h] = gelu_fast2(v[h]);
    }
    const f16x2 h0 = __builtin_convertvector(v[0], f16x2), h1 = __builtin_convertvector(v[1], f16x2);
    return __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
}

#ifdef MOGE_EXPERIMENTS
#include "../../tools/experiments/gemm_pp_exp.inc"     // 32x32x16-form kernels, 64-byte-row kernels: tools/kbench A-B builds only
#endif

// ---- 16x16x32 accumulators acc[i][j0 + jj] (128 rows x 64 columns of the wave's tile): row i*16 + (lane & 15), columns jj*16 + 4*(lane >> 4) .. +3 ----
template <int EPKX, int NJT = 8>
__device__ __forceinline__ void pp_epilogue16(const GemmArgs& g, f32x4 (&acc)[8][NJT], int j0, char* smem, int wave, int lane, int mw, int nw) {
    constexpr int EPK = EpkBase<EPKX>::K;
    constexpr bool FOLD = EpkBase<EPKX>::FOLD;
    constexpr int WROWS = 128;
    const int l15 = lane & 15, g4 = lane >> 4;
    char* R = smem + wave * (WROWS * 128);
    const int M = g.M;

    if constexpr (EPK == EPK_RESID16) {
        // two passes of 64 rows x 64 columns (256 B per staging row = the wave's 16 KiB)
        constexpr int PROWS = 64;
        const Resid16Bufs rb = pp_resid16_bufs(g);
        const bool fold = g.ln_part != nullptr;
        u32x4 x0[PROWS / 8], x1[PROWS / 8];
        pp_resid16_load<PROWS>(rb, lane, mw, nw, x0);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int n = nw + jj * 16 + 4 * g4;
            const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(g.gamma + n);
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][j0 + jj][e] = resid_term(gm[e], acc[i][j0 + jj][e], b[e]);
        }
        pp_resid16_stage<PROWS, NJT>(R, acc, j0, 0, lane);
        pp_resid16_add<PROWS>(R, lane, x0);
        pp_resid16_load<PROWS>(rb, lane, mw + PROWS, nw, x1);
        pp_resid16_store<PROWS>(rb, fold, lane, mw, nw, x0);
        pp_resid16_stage<PROWS, NJT>(R, acc, j0, PROWS / 16, lane);
        pp_resid16_add<PROWS>(R, lane, x1);
        pp_resid16_store<PROWS>(rb, fold, lane, mw + PROWS, nw, x1);
    } else if constexpr (EPK == EPK_RESID) {
        // two passes of 32 fp32 columns (128 B per staging row); pass 1's row loads are in flight before pass 0's stores are issued.
        // gamma (acc + bias) is applied IN PLACE to all accumulators first: a bias / gamma load issued after the first stores would sit
        // behind them in the in-order vmcnt queue
        auto stage = [&](int J) {
#pragma unroll
            for (int jh = 0; jh < 2; jh++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int row = i * 16 + l15;
                    *reinterpret_cast<f32x4*>(R + row * 128 + (((jh * 4 + g4) ^ (row & 7)) << 4)) = acc[i][j0 + J * 2 + jh];
                }
        };
        const ResidBufs rb = pp_resid_bufs(g);
        const bool fold = g.x16 != nullptr;
        f32x4 x0[WROWS / 8], x1[WROWS / 8];
        pp_resid_load<WROWS>(rb, lane, mw, nw, x0);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int n = nw + jj * 16 + 4 * g4;
            const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(g.gamma + n);
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][j0 + jj][e] = resid_term(gm[e], acc[i][j0 + jj][e], b[e]);
        }
        stage(0);
        pp_resid_add<WROWS>(R, lane, x0);
        pp_resid_load<WROWS>(rb, lane, mw, nw + 32, x1);
        pp_resid_store<WROWS>(rb, fold, lane, mw, nw, x0);
        stage(1);
        pp_resid_add<WROWS>(R, lane, x1);
        pp_resid_store<WROWS>(rb, fold, lane, mw, nw + 32, x1);
    } else {
        float scale = 1.f;
        if constexpr (EPK == EPK_QKV) scale = nw < g.D ? g.qscale : 1.f;
        float u[8], vv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { u[i] = 0.f; vv[i] = 0.f; }
        if constexpr (EPK == EPK_UV) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int m = mw + i * 16 + l15;
                m = m < M ? m : M - 1;
                const int x = m % g.pixW, y = (m / g.pixW) % g.pixH;
                u[i] = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, g.pixW, x);
                vv[i] = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, g.pixH, y);
            }
        }
        const bool has_bias = g.bias != nullptr;
        float mu[8], rs[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { mu[i] = 0.f; rs[i] = 1.f; }
        if constexpr (FOLD) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int m = mw + i * 16 + l15;
                m = m < M ? m : M - 1;
                const f32x2 t = *reinterpret_cast<const f32x2*>(g.ln_mr + 2 * (size_t)m);
                mu[i] = t[0]; rs[i] = t[1];
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int n = nw + jj * 16 + 4 * g4;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (has_bias) b = *reinterpret_cast<const f32x4*>(g.bias + n);
            f32x4 lc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (FOLD) lc = *reinterpret_cast<const f32x4*>(g.ln_c + n);
            f32x4 wu = {0.f, 0.f, 0.f, 0.f}, wv = wu;
            if constexpr (EPK == EPK_UV) {
                wu = *reinterpret_cast<const f32x4*>(g.uv.wu + n);
                wv = *reinterpret_cast<const f32x4*>(g.uv.wv + n);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int row = i * 16 + l15;
                const f16x4 hv = pp_quad_f16<EPK, FOLD>(acc[i][j0 + jj], b, scale, wu, u[i], wv, vv[i], mu[i], rs[i], lc);
                *reinterpret_cast<f16x4*>(R + row * 128 + ((((jj * 2 + (g4 >> 1)) ^ (row & 7)) << 4) | ((g4 & 1) << 3))) = hv;
            }
        }
        pp_store_rows<WROWS, EPK>(g, R, lane, mw, nw);
    }
}

#ifdef MOGE_EXPERIMENTS
#include "../../tools/experiments/gemm_pp4w16_exp.inc"     // 4-wave form (128 x 128 per wave): slower on every hot-path shape, tools/kbench A-B builds only
#endif

// ------------------------------------------------------------------------------------------------------------------------
// gemm_pp128 (A3 rings, 8 waves, ping-pong) with v_mfma_f32_16x16x32_f16: per phase 4 row blocks x 4 column blocks x 2 K-steps of 32
// = 32 MFMAs (16 cycles each), 8 + 8 fragment reads - the same LDS layout, DMA pieces, waits and barriers as the 32x32x16 kernel;
// only the fragment addressing (row = block*16 + (lane & 15), chunk = 4*kstep + (lane >> 4)) and the accumulator layout change.
// ------------------------------------------------------------------------------------------------------------------------
template <int EPK>
__global__ __launch_bounds__(512, 2) void gemm_pp128m16_kernel(const GemmArgs g) {
    constexpr int WN = 4, BM = 256, BN = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 3 * 32 KiB (A ring) + 2 * 32 KiB (W ring)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;

    const int nbn = g.N / BN;
    int wg;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    int bm, bn;
    {
        const int nbm = (g.M + BM - 1) / BM;
        const int grp_cols = 4, per_grp = nbm * grp_cols;
        const int cg = wg / per_grp, rem = wg - cg * per_grp;
        const int cols = min(grp_cols, nbn - cg * grp_cols);
        bm = rem / cols;
        bn = cg * grp_cols + (rem - bm * cols);
    }
    const int m0 = bm * BM, n0 = bn * BN;
    const int nkt = g.K >> 6;

    const int prow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) + (lane >> 4));
    const char* baseW = reinterpret_cast<const char*>(g.w) + (size_t)n0 * g.ldw * 2;
    const char* baseA = reinterpret_cast<const char*>(g.a) + (size_t)m0 * g.lda * 2;
    unsigned offW[4], offA[4];     // A: 0,1 = lo rows (8w, 128+8w)   2,3 = hi rows (64+8w, 192+8w)
    const int mlast = g.M - 1 - m0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        offW[k] = (unsigned)(((wave + 8 * k) * 8 + prow) * g.ldw * 2 + lchunk * 16);
        int arow = (k & 1) * 128 + (k >> 1) * 64 + wave * 8 + prow;
        arow = arow < mlast ? arow : mlast;
        offA[k] = (unsigned)(arow * g.lda * 2 + lchunk * 16);
    }
    auto issue_w = [&](int t) {
        char* base = smem + 98304 + (t & 1) * 32768;
        const char* gw = uniform_ptr(baseW + (size_t)t * 128);
#pragma unroll
        for (int kw = 0; kw < 4; kw++) __builtin_amdgcn_global_load_lds(PP_GPTR(gw + pp_opaque(offW[kw])), PP_LPTR(base + (wave + 8 * kw) * 1024), 16, 0, 0);
    };
    auto issue_a = [&](int t, int slot, int hi_rows) {
        char* base = smem + slot * 32768;
        const char* ga = uniform_ptr(baseA + (size_t)t * 128);
#pragma unroll
        for (int k = 0; k < 2; k++)
            __builtin_amdgcn_global_load_lds(PP_GPTR(ga + pp_opaque(offA[hi_rows * 2 + k])), PP_LPTR(base + (k * 128 + hi_rows * 64 + wave * 8) * 128), 16, 0, 0);
    };

    const int sx = (l15 >> 1) & 7;
    const int a_off = (wm * 128 + l15) * 128 + ((g4 ^ sx) << 4);             // K-step ks (32 halves): a_off ^ (ks * 64); 16-row block i: + i * 2048
    const int w_off = (wn * 64 + l15) * 128 + ((g4 ^ sx) << 4);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_w(0); issue_a(0, 0, 0); issue_a(0, 0, 1);
    if (nkt > 1) { issue_a(1, 1, 0); issue_a(1, 1, 1); issue_w(1); }
    if (nkt > 2) {
        issue_a(2, 2, 0);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else if (nkt > 1) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    u32x4 af[4][2], wf[4][2];
    int sa = 0;
    for (int t = 0; t < nkt; t++) {
        const char* sl = smem + sa * 32768;
        const char* slw = smem + 98304 + (t & 1) * 32768;
        const int sa2 = sa == 0 ? 2 : sa - 1;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            // ======== load segment ========
            if (half == 0) {
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) wf[j][ks] = *reinterpret_cast<const u32x4*>(slw + (w_off ^ (ks * 64)) + j * 2048);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) af[i][ks] = *reinterpret_cast<const u32x4*>(sl + (a_off ^ (ks * 64)) + (half * 4 + i) * 2048);
            if (half == 0) {
                if (t + 2 < nkt) issue_a(t + 2, sa2, 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                if (t + 3 < nkt) {
                    issue_w(t + 2);
                    issue_a(t + 3, sa, 0);
                    asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
                } else if (t + 2 < nkt) {
                    issue_w(t + 2);
                    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ======== compute segment ========
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) mma16<f16>(acc[half * 4 + i][j], wf[j][ks], af[i][ks]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            if (!(grp == 1 && half == 1 && t == nkt - 1)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        sa = sa == 2 ? 0 : sa + 1;
    }
    pp_epilogue16<EPK, 4>(g, acc, 0, smem, wave, lane, m0 + wm * 128, n0 + wn * 64);
}

// ---- the epilogue of the persistent kernel: staging region of SROWS (64 / 32) rows per wave, 128 / SROWS row passes ------------------
// Split in two so that the kernel can request the NEXT tile's first DMA pieces in between:
//   pp_epi_pre   every load the epilogue needs before its first store (bias / LayerScale / folded-LN vectors, the first residual rows).
//                Issued BEFORE the DMA prefetch: vmcnt retires in order, so a load issued behind the prefetch would be waited for
//                together with it (the compiler's s_waitcnt vmcnt(0) in front of the first use exposed the whole prefetch latency).
//   pp_epi_run   arithmetic, staging, stores.  All stores are raw buffer stores without branches (rows >= M are given an offset outside
//                the descriptor and dropped): the staging reads of a pass are issued together instead of one read-wait-store per row, and
//                the number of memory instructions behind the prefetch is a constant - the tile-head wait counts them (PP_TRAIL).
template <int EPKX, int SROWS> struct PpEpiPre {
    f32x4 bq[4], lq[4], wuq[4], wvq[4];
    float mu[8], rs[8], u[8], vv[8];
    f32x4 x0[SROWS / 8];
    ResidBufs rb;
    u32x4 h[256 / SROWS][SROWS / 16];   // EPK_RESID16: the wave's residual tile (128 rows x 64 fp16 columns = 64 registers), requested in two steps - pass 0 in
                                        // pp_epi_pre (into registers the A / W fragments have left), passes 1 ... 3 together once the LayerScale / bias vectors are
                                        // dead (pp_epi_run) - instead of one pass ahead: the load latency is exposed once per tile, not once per pass
    Resid16Bufs rb16;
    __amdgpu_buffer_rsrc_t ob;
    float scale;
    bool fold;
};
constexpr unsigned PP_OOB = 0xffffff00u;

template <int EPKX, int SROWS>
__device__ __forceinline__ void pp_epi_pre(const GemmArgs& g, PpEpiPre<EPKX, SROWS>& P, int lane, int mw, int nw) {
    constexpr int EPK = EpkBase<EPKX>::K;
    constexpr bool FOLD = EpkBase<EPKX>::FOLD;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int M = g.M;
    if constexpr (EPK == EPK_RESID16) {
        P.rb16 = pp_resid16_bufs(g);
        P.fold = g.ln_part != nullptr;
#pragma unroll
        for (int p = 0; p < 1; p++) pp_resid16_load<SROWS / 2>(P.rb16, lane, mw + p * (SROWS / 2), nw, P.h[p]);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int n = nw + jj * 16 + 4 * g4;
            P.bq[jj] = *reinterpret_cast<const f32x4*>(g.bias + n);
            P.lq[jj] = *reinterpret_cast<const f32x4*>(g.gamma + n);
        }
    } else if constexpr (EPK == EPK_RESID) {
        P.rb = pp_resid_bufs(g);
        P.fold = g.x16 != nullptr;
        pp_resid_load<SROWS>(P.rb, lane, mw, nw, P.x0);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int n = nw + jj * 16 + 4 * g4;
            P.bq[jj] = *reinterpret_cast<const f32x4*>(g.bias + n);
            P.lq[jj] = *reinterpret_cast<const f32x4*>(g.gamma + n);
        }
    } else {
        P.scale = 1.f;
        if constexpr (EPK == EPK_QKV) P.scale = nw < g.D ? g.qscale : 1.f;
#pragma unroll
        for (int i = 0; i < 8; i++) { P.u[i] = 0.f; P.vv[i] = 0.f; P.mu[i] = 0.f; P.rs[i] = 1.f; }
        if constexpr (EPK == EPK_UV) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int m = mw + i * 16 + l15;
                m = m < M ? m : M - 1;
                const int x = m % g.pixW, y = (m / g.pixW) % g.pixH;
                P.u[i] = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, g.pixW, x);
                P.vv[i] = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, g.pixH, y);
            }
        }
        if constexpr (FOLD) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int m = mw + i * 16 + l15;
                m = m < M ? m : M - 1;
                const f32x2 t = *reinterpret_cast<const f32x2*>(g.ln_mr + 2 * (size_t)m);
                P.mu[i] = t[0]; P.rs[i] = t[1];
            }
        }
        const bool has_bias = g.bias != nullptr;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int n = nw + jj * 16 + 4 * g4;
            P.bq[jj] = f32x4{0.f, 0.f, 0.f, 0.f}; P.lq[jj] = P.bq[jj]; P.wuq[jj] = P.bq[jj]; P.wvq[jj] = P.bq[jj];
            if (has_bias) P.bq[jj] = *reinterpret_cast<const f32x4*>(g.bias + n);
            if constexpr (FOLD) P.lq[jj] = *reinterpret_cast<const f32x4*>(g.ln_c + n);
            if constexpr (EPK == EPK_UV) {
                P.wuq[jj] = *reinterpret_cast<const f32x4*>(g.uv.wu + n);
                P.wvq[jj] = *reinterpret_cast<const f32x4*>(g.uv.wv + n);
            }
        }
        // output descriptor: exactly the bytes the GEMM may write (pp_persistent_ok() has checked they fit 32-bit offsets)
        if constexpr (EPK == EPK_QKV) {
            const int which = nw / g.D;
            P.ob = __builtin_amdgcn_make_buffer_rsrc(which == 0 ? g.q : (which == 1 ? g.k : g.vT), 0, (int)((unsigned)M * (unsigned)g.nh * 128u), 0x00020000);
        } else if constexpr (EPK == EPK_CONVT) {
            P.ob = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((unsigned)M * (unsigned)g.Cout * 8u), 0x00020000);
        } else {
            P.ob = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((unsigned)M * (unsigned)g.ldc * 2u), 0x00020000);
        }
    }
}

// staging rows (64 f16 columns, 16-byte chunks swizzled by row & 7) -> global memory, SROWS / 8 unconditional 16-byte buffer stores per lane
template <int SROWS, int EPK>
__device__ __forceinline__ void pp_store_rows_buf(const GemmArgs& g, __amdgpu_buffer_rsrc_t ob, const char* R, int lane, int mw, int nw) {
    const int rr = lane >> 3, cc = lane & 7;
    const int M = g.M;
    const int mfirst = mw + rr;
    u32x4 v[SROWS / 8];
#pragma unroll
    for (int it = 0; it < SROWS / 8; it++) {
        const int row = it * 8 + rr;
        v[it] = *reinterpret_cast<const u32x4*>(R + row * 128 + ((cc ^ (row & 7)) << 4));
    }
    if constexpr (EPK == EPK_QKV) {
        const int which = nw / g.D;
        const int head = (nw - which * g.D) >> 6;
        const int Ntok = g.Ntok;
        const unsigned base = (unsigned)head * (unsigned)Ntok * 128u + (unsigned)cc * 16u;
        const unsigned bstride = (unsigned)g.nh * (unsigned)Ntok * 128u;
        int b0 = mfirst / Ntok;
        int t0 = mfirst - b0 * Ntok;
#pragma unroll
        for (int it = 0; it < SROWS / 8; it++) {
            unsigned off = base + (unsigned)b0 * bstride + (unsigned)t0 * 128u;
            if (mfirst + it * 8 >= M) off = PP_OOB;
            t0 += 8;
            if (t0 >= Ntok) { t0 -= Ntok; b0 += 1; }
            __builtin_amdgcn_raw_buffer_store_b128(v[it], ob, (int)off, 0, 0);
        }
    } else if constexpr (EPK == EPK_CONVT) {
        const int Cout = g.Cout, pixW = g.pixW, pixH = g.pixH;
        const int qd = nw / Cout;
        const int co0 = nw - qd * Cout, dy = qd >> 1, dx = qd & 1;
        int px = mfirst % pixW;
        const int t = mfirst / pixW;
        int py = t % pixH, pb = t / pixH;
#pragma unroll
        for (int it = 0; it < SROWS / 8; it++) {
            unsigned off = ((((unsigned)pb * 2u * pixH + 2u * py + dy) * (2u * pixW)) + 2u * px + dx) * ((unsigned)Cout * 2u) + (unsigned)(co0 + cc * 8) * 2u;
            if (mfirst + it * 8 >= M) off = PP_OOB;
            px += 8;
            if (px >= pixW) { px -= pixW; py += 1; if (py >= pixH) { py = 0; pb += 1; } }
            __builtin_amdgcn_raw_buffer_store_b128(v[it], ob, (int)off, 0, 0);
        }
    } else {
        const unsigned ldc2 = (unsigned)g.ldc * 2u;
        const unsigned off0 = (unsigned)mfirst * ldc2 + (unsigned)(nw + cc * 8) * 2u;
#pragma unroll
        for (int it = 0; it < SROWS / 8; it++)
            __builtin_amdgcn_raw_buffer_store_b128(v[it], ob, (int)(mfirst + it * 8 < M ? off0 + (unsigned)it * 8u * ldc2 : PP_OOB), 0, 0);
    }
}

// mid(): called once, at the first point behind which the epilogue has no load left that the COMPILER waits for (it places s_waitcnt vmcnt(0)
// in front of the first use of any ordinary load while LDS-DMA is in flight): the kernel requests the next tile's first pieces there.
// Behind it follow PpTrail<> unconditional stores per lane.
// (a LOWER bound would be safe - the head wait then also covers some stores - and costly: it makes every tile head wait for store
//  acknowledgements.  The RESID flavours therefore issue their optional outputs - fp16 copy, LN partials - UNCONDITIONALLY in this kernel: a
//  null output is an empty buffer descriptor, its stores are dropped by the bounds check, and the count is exact: x + x16 + partials.)
template <int EPKX, int SROWS> struct PpTrail {
    static constexpr int N = EpkBase<EPKX>::K == EPK_RESID ? 3 * (SROWS / 8) : (EpkBase<EPKX>::K == EPK_RESID16 ? 2 * (SROWS / 16) : 16);
};
template <int EPKX, int SROWS, class Mid>
__device__ __forceinline__ void pp_epi_run(const GemmArgs& g, f32x4 (&acc)[8][4], PpEpiPre<EPKX, SROWS>& P, char* R, int lane, int mw, int nw, Mid mid) {
    constexpr int EPK = EpkBase<EPKX>::K;
    constexpr bool FOLD = EpkBase<EPKX>::FOLD;
    constexpr int NI = SROWS / 16, NP = 128 / SROWS;
    const int l15 = lane & 15, g4 = lane >> 4;

    if constexpr (EPK == EPK_RESID16) {
        // 128 / PROWS passes of PROWS = SROWS / 2 rows x all 64 columns (256-byte staging rows fill the wave's SROWS x 128 B region)
        constexpr int PROWS = SROWS / 2, NPASS = 128 / PROWS;
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][jj][e] = resid_term(P.lq[jj][e], acc[i][jj][e], P.bq[jj][e]);
        __builtin_amdgcn_sched_barrier(0);                     // (not above the loop: with lq / bq still live the second half does not fit 256 registers)
#pragma unroll
        for (int p = 1; p < NPASS; p++) pp_resid16_load<PROWS>(P.rb16, lane, mw + p * PROWS, nw, P.h[p]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            u32x4 (&xc)[PROWS / 8] = P.h[p];
            pp_resid16_stage<PROWS, 4>(R, acc, 0, p * (PROWS / 16), lane);
            pp_resid16_add<PROWS>(R, lane, xc);
            if (p + 1 == NPASS) mid();
            pp_resid16_store<PROWS>(P.rb16, true, lane, mw + p * PROWS, nw, xc);       // (statistics stores always issued: see PpTrail)
        }
    } else if constexpr (EPK == EPK_RESID) {
        f32x4 x1[SROWS / 8];
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][jj][e] = resid_term(P.lq[jj][e], acc[i][jj][e], P.bq[jj][e]);
#pragma unroll
        for (int p = 0; p < 2 * NP; p++) {                   // pass p: rows (p >> 1) * SROWS .., columns (p & 1) * 32 ..
            const int ih = p >> 1, J = p & 1;
            f32x4 (&xc)[SROWS / 8] = (p & 1) ? x1 : P.x0;
            f32x4 (&xn)[SROWS / 8] = (p & 1) ? P.x0 : x1;
#pragma unroll
            for (int jh = 0; jh < 2; jh++)
#pragma unroll
                for (int ii = 0; ii < NI; ii++) {
                    const int row = ii * 16 + l15;
                    *reinterpret_cast<f32x4*>(R + row * 128 + (((jh * 4 + g4) ^ (row & 7)) << 4)) = acc[ih * NI + ii][J * 2 + jh];
                }
            pp_resid_add<SROWS>(R, lane, xc);
            if (p + 1 < 2 * NP) pp_resid_load<SROWS>(P.rb, lane, mw + ((p + 1) >> 1) * SROWS, nw + ((p + 1) & 1) * 32, xn);
            else mid();
            pp_resid_store<SROWS>(P.rb, true, lane, mw + ih * SROWS, nw + J * 32, xc);        // (fp16 copy / statistics always issued: see PpTrail)
        }
    } else {
#pragma unroll
        for (int ih = 0; ih < NP; ih++) {
#pragma unroll
            for (int jj = 0; jj < 4; jj++)
#pragma unroll
                for (int ii = 0; ii < NI; ii++) {
                    const int row = ii * 16 + l15, i = ih * NI + ii;
                    const f16x4 hv = pp_quad_f16<EPK, FOLD>(acc[i][jj], P.bq[jj], P.scale, P.wuq[jj], P.u[i], P.wvq[jj], P.vv[i], P.mu[i], P.rs[i], P.lq[jj]);
                    *reinterpret_cast<f16x4*>(R + row * 128 + ((((jj * 2 + (g4 >> 1)) ^ (row & 7)) << 4) | ((g4 & 1) << 3))) = hv;
                }
            if (ih == 0) mid();
            pp_store_rows_buf<SROWS, EPK>(g, P.ob, R, lane, mw + ih * SROWS, nw);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// gemm_pp128p_kernel: gemm_pp128m16_kernel as a PERSISTENT kernel - one workgroup per CU walks a list of output tiles.
// Why: a K = 1024 tile is a 23 us main loop (16 ring steps) + ~8 us of workgroup launch, cold prologue (144 KiB of DMA before the first
// MFMA) and epilogue; with one 160 KiB workgroup per CU nothing of that overlaps (the next workgroup cannot start before this one has
// left).  Here the next tile's first ring pieces (A step 0, W steps 0 and 1: 96 KiB) are requested BEFORE the epilogue of the current
// tile and land while it runs; the epilogue stages through the other 64 KiB of the ring (A slots 1 and 2, 8 KiB per wave, 64-row passes).
// Tiles: each XCD owns a contiguous range of the (grouped) tile order, its workgroups (blockIdx & 7 = XCD) take consecutive tiles of it,
// so the A rows / W columns a round works on are shared in that XCD's L2 as in the one-tile-per-workgroup kernel.
// Queue order of a tile's DMA (MODE 3, the product): A0 W0 W1 | epilogue loads, stores | A1 | A2 W2 | A3 W3 ... : every K-tile's read segments request
// A(t+2) (4 pieces per wave, first segment) and W(t+2) (4 pieces, second segment); the steady-state wait at the end of the second segment is
// vmcnt(8) (A(t+2) and W(t+2) stay in flight, everything up to W(t+1) has landed), the tile-head wait vmcnt(4) (A1 stays in flight).
// (Rounds 2-3 requested A_hi(t+2) | W(t+2) A_lo(t+3) = 2 + 6 pieces: the second read segment - 8 fragment reads, 6 DMA issues, the counted
//  wait - overran the partner's 32 MFMAs; 4 + 4 is +3.3 ... +4.9 % at K = 1024 and +1.7 ... +2.6 % at K = 4096, profiles/r04n-r04r.)
// ------------------------------------------------------------------------------------------------------------------------
// MRG (round 3 experiment, instantiated in --experiments builds only; measured 10-17 % SLOWER, see launch_pp_any): ONE MFMA segment of 64 per
// K-tile and wave group instead of two of 32 - half the hand-overs between the two wave groups of a SIMD (each costs ~90 clocks of barrier +
// restart, conv_rb.hip's step timeline: profiles/r03j_kbench_rb_step_timeline.log) - at the price of a W ring that is only 0.75 K-tiles ahead.  The A rows
// 64-127 of the wave ("hi") are read from LDS INSIDE the segment, into the registers of the "lo" fragments as those die, so the register
// count does not change.  The DMA schedule is then tied to the global intervals (both groups issue at the start of an even interval - group 0
// is in its read segment, group 1 in its MFMA segment - and wait at the interval ends), see the loop.
constexpr int PP_WX_DEFAULT = 0;      // (measured slower in the production step: see launch_pp_any)
template <int EPK, int SROWS, int MODE>
__global__ __launch_bounds__(512, 2) void gemm_pp128p_kernel(const GemmArgs g) {
    constexpr bool MRG = MODE == 1;
    constexpr bool WDMA_M = MODE == 2;         // W(t+2) requested at the head of the SECOND MFMA segment of K-tile t instead of in the read segment in front of it
    // MODE 3 (round 4): BOTH halves of A(t+2) are requested in the FIRST read segment of K-tile t (4 pieces per wave), W(t+2) alone in the second
    // (4 pieces) - instead of 2 + 6: the second read segment (8 fragment reads + 6 DMA issues) was the one that overran the partner's 32 MFMAs.
    // Queue per K-tile: [L_a(t): A(t+2)] [L_b(t): W(t+2)], wait at the end of L_b(t): vmcnt(8) = everything up to W(t+1) has landed.
    constexpr bool ALO_A = MODE == 3 || MODE == 5 || MODE == 7;
    // MODE 7 (round 6) = MODE 3 with the W stream CONTINUOUS across the tile boundary: the last two K-tiles of a tile, whose second read segments have no W(t + 2) of
    // their own to request, request the NEXT tile's W(0) and W(1) there (into the W slots that are dead by then, exactly the steady-state rule) - 8 of the 12 DMA
    // pieces per wave leave the epilogue's prefetch, where 256 workgroups issue them together with the store burst (230-700 clocks per piece against 13-20 in the
    // main loop: EXPERIMENTS R6.3c).  Needs an even number of K-tiles (slot parity); the next tile's descriptor is set up at the head of K-tile nkt - 2.
    constexpr bool WX = MODE == 7;
    // MODE 6: "6 + 2" - as MODE 3, and the second half of W(t+1) is requested at the head of L_a(t) instead of in L_b(t-1):
    // [L_a(t): W(t+1) rows 128-255, A(t+2)] [L_b(t): W(t+2) rows 0-127]; wait at the end of L_b(t): vmcnt(6)
    constexpr bool W_SPLIT = MODE == 6;
    constexpr bool DMA_FIRST = MODE == 5;      // MODE 5 = MODE 3 with the DMA pieces of a read segment issued BEFORE its fragment reads
    // MODE 4: 3 + 5 - the first read segment carries 16 fragment reads, the second 8, so one more piece goes to the second:
    // [L_a(t): A_hi(t+2), A_lo(t+2) rows 0-63] [L_b(t): W(t+2), A_lo(t+3) rows 128-191]; wait at the end of L_b(t): vmcnt(9)
    constexpr bool SPLIT35 = MODE == 4;
    constexpr int WN = 4, BM = 256, BN = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 3 * 32 KiB (A ring) + 2 * 32 KiB (W ring)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;

    const int nbn = g.N / BN, nbm = (g.M + BM - 1) / BM;
    int li, cnt, start, wgs_x;                               // this workgroup's position in its XCD's tile range
    {
        const int ntiles = nbm * nbn, nwg = gridDim.x;
        const int nx = nwg < 8 ? nwg : 8, xcd = blockIdx.x % nx;          // (fewer than 8 workgroups: as many ranges as workgroups)
        const int q = ntiles / nx, r = ntiles % nx;
        cnt = q + (xcd < r ? 1 : 0);
        start = xcd * q + min(xcd, r);
        wgs_x = (nwg - xcd + nx - 1) / nx;
        li = blockIdx.x / nx;
    }
    if (li >= cnt) return;
    const int nkt = g.K >> 6;
#ifdef MOGE_EXPERIMENTS
    // energy / time attribution (tools/energy.sh with PP_ABL): pieces of the kernel switched off at run time - wrong results, same control flow
    const int abl = g.abl;
#else
    constexpr int abl = 0;
#endif

    const int prow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) + (lane >> 4));
    unsigned offW[4], offA[4];     // A: 0,1 = lo rows (8w, 128+8w)   2,3 = hi rows (64+8w, 192+8w)
#pragma unroll
    for (int k = 0; k < 4; k++) offW[k] = (unsigned)(((wave + 8 * k) * 8 + prow) * g.ldw * 2 + lchunk * 16);
    const char* baseW;
    const char* baseA;
    int m0n, n0n;
    auto setup = [&](int idx) {                               // tile idx of the grouped order (4 tile columns per group, rows inside)
        const int wg = start + idx;
        const int grp_cols = g.dbg > 0 ? g.dbg : 8, per_grp = nbm * grp_cols;
        const int cg = wg / per_grp, rem = wg - cg * per_grp;
        const int cols = min(grp_cols, nbn - cg * grp_cols);
        const int bm = rem / cols;
        const int bn = cg * grp_cols + (rem - bm * cols);
        m0n = bm * BM; n0n = bn * BN;
        baseW = reinterpret_cast<const char*>(g.w) + (size_t)n0n * g.ldw * 2;
        baseA = reinterpret_cast<const char*>(g.a) + (size_t)m0n * g.lda * 2;
        const int mlast = g.M - 1 - m0n;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int arow = (k & 1) * 128 + (k >> 1) * 64 + wave * 8 + prow;
            arow = arow < mlast ? arow : mlast;
            offA[k] = (unsigned)(arow * g.lda * 2 + lchunk * 16);
        }
    };
    auto issue_w = [&](int t) {
        if ((abl & 1) && t >= 2) return;
        char* base = smem + 98304 + (t & 1) * 32768;
        const char* gw = uniform_ptr(baseW + (size_t)t * 128);
#pragma unroll
        for (int kw = 0; kw < 4; kw++) __builtin_amdgcn_global_load_lds(PP_GPTR(gw + pp_opaque(offW[kw])), PP_LPTR(base + (wave + 8 * kw) * 1024), 16, 0, 0);
    };
    auto issue_a = [&](int t, int slot, int hi_rows) {
        if ((abl & 1) && t >= 3) return;
        char* base = smem + slot * 32768;
        const char* ga = uniform_ptr(baseA + (size_t)t * 128);
#pragma unroll
        for (int k = 0; k < 2; k++)
            __builtin_amdgcn_global_load_lds(PP_GPTR(ga + pp_opaque(offA[hi_rows * 2 + k])), PP_LPTR(base + (k * 128 + hi_rows * 64 + wave * 8) * 128), 16, 0, 0);
    };
    auto issue_w2 = [&](int t, int part) {                    // half of a W K-tile: part 0 = pieces kw 0, 1 (rows 0-127), part 1 = kw 2, 3 (rows 128-255)
        char* base = smem + 98304 + (t & 1) * 32768;
        const char* gw = uniform_ptr(baseW + (size_t)t * 128);
#pragma unroll
        for (int kw = 2 * part; kw < 2 * part + 2; kw++) __builtin_amdgcn_global_load_lds(PP_GPTR(gw + pp_opaque(offW[kw])), PP_LPTR(base + (wave + 8 * kw) * 1024), 16, 0, 0);
    };
    auto issue_a1 = [&](int t, int slot, int k) {             // ONE piece of the "lo" rows: k = 0 rows 8w (wave group 0's), k = 1 rows 128 + 8w (group 1's)
        char* base = smem + slot * 32768;
        const char* ga = uniform_ptr(baseA + (size_t)t * 128);
        __builtin_amdgcn_global_load_lds(PP_GPTR(ga + pp_opaque(offA[k])), PP_LPTR(base + (k * 128 + wave * 8) * 128), 16, 0, 0);
    };
    auto prefetch = [&]() {                                   // the pieces that do not touch the staging region (A slots 1, 2)
        // straight-line (K >= 192 is a launch condition): a branch in here makes the compiler wait vmcnt(0) for the epilogue's loads behind it
        issue_a(0, 0, 0); issue_a(0, 0, 1);
        issue_w(0);
        issue_w(1);
    };
    auto prefetch_a = [&]() { issue_a(0, 0, 0); issue_a(0, 0, 1); };      // MODE 7: W(0), W(1) of the next tile were requested by the last two K-tiles

    const int sx = (l15 >> 1) & 7;
    const int a_off = (wm * 128 + l15) * 128 + ((g4 ^ sx) << 4);
    const int w_off = (wn * 64 + l15) * 128 + ((g4 ^ sx) << 4);
    char* R = smem + 32768 + wave * (SROWS * 128);

#ifdef MOGE_EXPERIMENTS
    // tools/kbench KB_TS: s_memtime stamps (100 MHz) of waves 0 and 4 of one workgroup over its first 6 tiles
    unsigned long long* ts = (g.dbg_ts && blockIdx.x == 8 && lane == 0 && (wave & 3) == 0) ? g.dbg_ts + (wave >> 2) * 48 : nullptr;
    int ts_tile = 0;
#define PP_STAMP(k) do { if (ts && ts_tile < 6) ts[ts_tile * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PP_STAMP(k) do { } while (0)
#endif
    setup(li);
    prefetch();
#ifdef MOGE_EXPERIMENTS
    // STAGGER (round 5, tools/kbench KB_STAG / PP_STAGGER; not in the product).  All 256 workgroups of a persistent launch start together and stay in lockstep, so
    // their epilogues arrive as ONE burst per round: 32-70 MB of stores (+ residual reads) at once = 4.5-6.7 TB/s, and the next tile's operand requests queue behind
    // it - ablating the epilogue recovers 19-27 % of the K = 1024 launches (profiles/r05l_energy_attribution_*.log) although its arithmetic is a few k clocks.  With
    // `stagger` = G > 1 the workgroups of an XCD start in G phase groups spread over one tile time (at most stagger_clk clocks).  Measured: qkv +2.6 %, proj +5-10 %,
    // fc1 +1.5 %, fc2 0 in kbench, GEMM class 63.2 -> 61.9 ms in the single-stream profile - and 246 -> 244 img/s in the production two-stream step, B = 1 p50 6.1 ->
    // 6.5 ms (profiles/r05n_*): a sleeping workgroup holds its CU, and the other stream already fills the ragged ends the stagger creates.
    if (g.stagger > 1) {
        const int ph = li % g.stagger;
        int t_spread = nkt * 2900 + 9000;
        t_spread = t_spread < g.stagger_clk ? t_spread : g.stagger_clk;
        for (int n = ph * t_spread / g.stagger / 8128; n > 0; n--) __builtin_amdgcn_s_sleep(127);      // s_sleep 127 = 64 x 127 clocks
    }
#endif
    bool first = true;
    for (;;) {
        const int m0 = m0n, n0 = n0n;
        const int li_next = li + wgs_x;
        const bool more = li_next < cnt;
        PP_STAMP(0);
        // (no zero fill in the product form: the first K-tile is peeled and its first 64 MFMAs take the constant 0 as their C operand - 128 v_mov per
        //  wave and tile less, with the matrix pipe idle while they would run)
        f32x4 acc[8][4];
        if constexpr (MRG) {
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // tile head: the prefetched pieces (A0 W0 W1) are OLDER than the previous epilogue's last PP_TRAIL memory instructions (stores);
        // those may stay in flight
        constexpr int PP_TRAIL = PpTrail<EPK, SROWS>::N, POST = (ALO_A || W_SPLIT) ? 4 : (SPLIT35 ? 5 : 6);
        issue_a(1, 1, 0); issue_a(1, 1, 1);
        if constexpr (SPLIT35) issue_a1(2, 2, 1);
        else if constexpr (!ALO_A && !W_SPLIT) issue_a(2, 2, 0);
        if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(POST) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(POST + PP_TRAIL) : "memory");
        first = false;
        __builtin_amdgcn_s_barrier();
        if (!MRG && grp == 1) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(1);

        u32x4 af[4][2], wf[4][2];
        int sa = 0;
        if constexpr (MRG) {
            // Interval 2t:   group 0 reads the fragments of K-tile t (W, A rows 0-63), group 1 runs the 64 MFMAs of K-tile t - 1;
            // interval 2t+1: group 0 runs the 64 MFMAs of K-tile t, group 1 reads the fragments of K-tile t.  One barrier ends every interval.
            // DMA (every wave, whatever its group is doing): at the START of interval 2t  W(t+1) A_hi(t+1) A_lo(t+2)  [their buffers: W(t-1) and
            // A_lo(t-1) were last read in interval 2t-1, A_hi(t-2) in interval 2t-2];  at the END of interval 2t: A_hi(t) has landed (vmcnt: it
            // leaves A_lo(t+1) and this interval's pieces in flight);  at the END of interval 2t+1: W(t+1) and A_lo(t+1) have landed (A_hi(t+1),
            // A_lo(t+2) stay in flight).  The tile head has requested A_lo(1) A_hi(1) A_lo(2) (and W(1) landed with W(0)).
            auto read_frags = [&](int t, int slot) {
                const char* sl = smem + slot * 32768;
                const char* slw = smem + 98304 + (t & 1) * 32768;
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) wf[j][ks] = *reinterpret_cast<const u32x4*>(slw + (w_off ^ (ks * 64)) + j * 2048);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) af[i][ks] = *reinterpret_cast<const u32x4*>(sl + (a_off ^ (ks * 64)) + i * 2048);
            };
            auto mfma64 = [&](int slot, bool last = false) {    // K-tile whose A tile sits in `slot`; W and A-lo fragments are in wf / af
                const char* sl = smem + slot * 32768;
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                u32x4 ah[4][2];
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) mma16<f16>(acc[i][j], wf[j][ks], af[i][ks]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; i++) ah[i][ks] = *reinterpret_cast<const u32x4*>(sl + (a_off ^ (ks * 64)) + (4 + i) * 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (last) {
                    // group 1's last K-tile runs while group 0 is already in its epilogue, which stages through A slots 1 and 2: group 0 waits
                    // (behind its epilogue's loads) for this barrier = group 1's last LDS read of the ring
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) mma16<f16>(acc[4 + i][j], wf[j][ks], ah[i][ks]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto end_interval = [&]() {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            // (one straight-line loop PER GROUP: with the group test inside the loop the accumulators meet in 128 phi nodes at every join and
            //  the allocator spills hundreds of registers - scratch traffic that would also break the counted vmcnt waits)
            auto dma_even = [&](int t, int sa1, int sa2) {      // start of interval 2t
                if (t + 1 < nkt) { issue_w(t + 1); issue_a(t + 1, sa1, 1); }
                if (t + 2 < nkt) issue_a(t + 2, sa2, 0);
            };
            auto wait_even = [&](int t) {                       // end of interval 2t (t >= 1): A_hi(t) has landed
                if (t + 2 < nkt) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
                else if (t + 1 < nkt) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            };
            auto wait_odd = [&](int t) {                        // end of interval 2t+1: W(t+1), A_lo(t+1) have landed
                if (t + 2 < nkt) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else if (t + 1 < nkt) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            };
            if (grp == 0) {
                for (int t = 0; t < nkt; t++) {
                    const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa == 0 ? 2 : sa - 1;        // slots of K-tiles t+1 and t+2 (= t-1)
                    if (t >= 1) dma_even(t, sa1, sa2);
                    read_frags(t, sa);
                    if (t >= 1) wait_even(t);
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    end_interval();
                    mfma64(sa);
                    wait_odd(t);
                    end_interval();
                    sa = sa1;
                }
            } else {
                // t = 0: nothing to compute in the even interval
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                end_interval();
                read_frags(0, 0);
                wait_odd(0);
                end_interval();
                sa = 1;
                for (int t = 1; t < nkt; t++) {
                    const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa == 0 ? 2 : sa - 1;
                    dma_even(t, sa1, sa2);
                    mfma64(sa2);                               // K-tile t - 1
                    wait_even(t);
                    end_interval();
                    read_frags(t, sa);
                    wait_odd(t);
                    end_interval();
                    sa = sa1;
                }
                mfma64(sa == 0 ? 2 : sa - 1, true);            // the last K-tile (group 0 is in its epilogue)
            }
        } else {
        auto ktile = [&](int t, auto first_tile) {
            constexpr bool FIRST = decltype(first_tile)::value;
            const char* sl = smem + sa * 32768;
            const char* slw = smem + 98304 + (t & 1) * 32768;
            const int sa2 = sa == 0 ? 2 : sa - 1;
            if constexpr (WX) {
                // this tile's last DMA request was W(nkt - 1), in K-tile nkt - 3: its descriptor is dead, the next tile's replaces it (after the last tile: this tile again, never read)
                if (t == nkt - 2) setup(more ? li_next : li);
            }
#pragma unroll
            for (int half = 0; half < 2; half++) {
                if constexpr (DMA_FIRST) {
                    if (t + 2 < nkt) {
                        if (half == 0) { issue_a(t + 2, sa2, 0); issue_a(t + 2, sa2, 1); }
                        else issue_w(t + 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!(abl & 2) || t == 0) {
                if (half == 0) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int ks = 0; ks < 2; ks++) wf[j][ks] = *reinterpret_cast<const u32x4*>(slw + (w_off ^ (ks * 64)) + j * 2048);
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) af[i][ks] = *reinterpret_cast<const u32x4*>(sl + (a_off ^ (ks * 64)) + (half * 4 + i) * 2048);
                }
                if (half == 0) {
                    if constexpr (W_SPLIT) {
                        if (t >= 1 && t + 1 < nkt) issue_w2(t + 1, 1);
                        if (t + 2 < nkt) { issue_a(t + 2, sa2, 0); issue_a(t + 2, sa2, 1); }
                    } else
                    if (!DMA_FIRST && t + 2 < nkt) {
                        if constexpr (ALO_A) issue_a(t + 2, sa2, 0);
                        issue_a(t + 2, sa2, 1);
                        if constexpr (SPLIT35) issue_a1(t + 2, sa2, 0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
                    if constexpr (W_SPLIT) {
                        if (t + 2 < nkt) { issue_w2(t + 2, 0); asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); }
                        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    } else
                    if constexpr (SPLIT35) {
                        if (t + 3 < nkt) { issue_w(t + 2); issue_a1(t + 3, sa, 1); asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory"); }
                        else if (t + 2 < nkt) { issue_w(t + 2); asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); }
                        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    } else
                    if constexpr (WX) {
                        // queue at the end of this segment: [A(t+2)] W(t+2) while t + 2 < nkt; then W_next(0) (K-tile nkt - 2) and W_next(0) W_next(1) (K-tile nkt - 1):
                        // everything of THIS tile has landed, the next tile's pieces stay in flight
                        if (t + 2 < nkt) { issue_w(t + 2); asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); }
                        else if (t + 2 == nkt) { issue_w(0); asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); }
                        else { issue_w(1); asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); }
                    } else
                    if constexpr (ALO_A) {
                        if (t + 2 < nkt) { if constexpr (!DMA_FIRST) issue_w(t + 2); asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); }
                        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    } else
                    if constexpr (WDMA_M) {
                        // queue (oldest first): A_lo(t+2) | W(t+1) [head of M_b(t-1)] | A_hi(t+2) [L_a(t)] | A_lo(t+3) [here]: W(t+1) must have landed
                        if (t + 3 < nkt) { issue_a(t + 3, sa, 0); asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); }
                        else if (t + 2 < nkt) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    } else
                    if (t + 3 < nkt) {
                        issue_w(t + 2);
                        issue_a(t + 3, sa, 0);
                        asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
                    } else if (t + 2 < nkt) {
                        issue_w(t + 2);
                        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    }
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                if constexpr (WDMA_M) {
                    if (half == 1 && t + 2 < nkt) issue_w(t + 2);          // W(t)'s slot: read by both groups in their L_a(t), two barriers ago at the latest
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!(abl & 8) || FIRST) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (FIRST && ks == 0) acc[half * 4 + i][j] = mma16_first<f16>(wf[j][ks], af[i][ks]);
                            else mma16<f16>(acc[half * 4 + i][j], wf[j][ks], af[i][ks]);
                        }
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" ::: "memory");
                if (!(grp == 1 && half == 1 && t == nkt - 1)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            sa = sa == 2 ? 0 : sa + 1;
        };
        ktile(0, std::true_type{});
        for (int t = 1; t < nkt; t++) ktile(t, std::false_type{});
        }
        // every LDS read of the ring is complete (the other group's last load segment ended before the barrier this wave has passed)
        PP_STAMP(2);
        PpEpiPre<EPK, SROWS> pre;
        pp_epi_pre<EPK, SROWS>(g, pre, lane, m0 + wm * 128, n0 + wn * 64);
        if constexpr (MRG) {
            if (grp == 0) {                                  // pairs with the barrier inside group 1's last MFMA segment (see mfma64)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        li = li_next;
        PP_STAMP(3);
        auto mid = [&]() {
            // every load of the epilogue has been issued long ago: tell the compiler they are complete (a real S_WAITCNT it accounts for: vmcnt(0), other
            // counters untouched; some results are first USED later, and its own wait there would be vmcnt(0) behind the DMA), then
            // request the next tile's first pieces - UNCONDITIONALLY (after the last tile: this tile's own pieces again, never read): a branch
            // here would merge into a conservative wait as well
            PP_STAMP(6);
            if constexpr (!WX) setup(more ? li : li - wgs_x);
            if constexpr (EpkBase<EPK>::K != EPK_RESID && EpkBase<EPK>::K != EPK_RESID16) __builtin_amdgcn_s_waitcnt(0x0F70);      // (RESID: the last row loads were waited for by the add in front)
            if constexpr (WX) prefetch_a(); else prefetch();
            asm volatile("" ::: "memory");
            PP_STAMP(7);
        };
        if (abl & 4) {                                       // no epilogue: the accumulators stay alive, the next tile's prefetch is still requested
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) asm volatile("" ::"v"(acc[i][j]));
            mid();
        } else
        pp_epi_run<EPK, SROWS>(g, acc, pre, R, lane, m0 + wm * 128, n0 + wn * 64, mid);
        PP_STAMP(4);
#ifdef MOGE_EXPERIMENTS
        if (ts && ts_tile < 6) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PP_STAMP(5); }      // store drain (perturbs: only the stamped waves wait)
        ts_tile++;
#endif
        if (!more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; }      // no DMA may land in this CU's LDS after the workgroup has left
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // all staging regions read: A slots 1 and 2 may be refilled
        asm volatile("" ::: "memory");
    }
#undef PP_STAMP
}

static int pp_num_cus() { return pp_device_cus(); }      // per device (common.h)

// the persistent kernel addresses its outputs through buffer descriptors with 32-bit byte offsets
static bool pp_persistent_ok(const GemmArgs& g) {
    if (g.K < 192) return false;                             // three ring steps: the kernel's prologue is written without branches
    const unsigned long long lim = 0xfffff000ull;
    const unsigned long long M = (unsigned long long)g.M + 256;
    switch (g.epi) {
    case EPI_RESID: return M * g.ldc * 4 < lim;
    case EPI_QKV: return M * g.nh * 128 < lim;
    case EPI_CONVT: return M * g.Cout * 8 < lim;
    default: return M * g.ldc * 2 < lim;
    }
}

template <int EPK, int SROWS, int MODE = 0>
static int launch_pp128p(const GemmArgs& g, hipStream_t st) {
    constexpr int smem = 163840;
    constexpr auto kern = gemm_pp128p_kernel<EPK, SROWS, MODE>;
    if (int rc = set_dyn_lds<kern>(smem)) return rc;
    const long ntiles = (long)((g.M + 255) / 256) * (g.N / 256);
    long cap = moge_tune_get("PP_GRID", 0);                  // tests: a small grid makes small problems walk many tiles per workgroup
    if (cap <= 0) cap = pp_num_cus();
    const long grid = ntiles < cap ? ntiles : cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, g);
    return (int)hipGetLastError();
}

template <int EPK>
static int launch_pp128m16(const GemmArgs& g, hipStream_t st) {
    constexpr int smem = 163840;
    constexpr auto kern = gemm_pp128m16_kernel<EPK>;
    if (int rc = set_dyn_lds<kern>(smem)) return rc;
    const long nbm = (g.M + 255) / 256, nbn = g.N / 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(512), smem, st, g);
    return (int)hipGetLastError();
}

#ifdef MOGE_EXPERIMENTS
template <int EPK, int A3>
static int launch_pp128_cfg(const GemmArgs& g, hipStream_t st) {
    constexpr int smem = A3 ? 163840 : 2 * 65536;
    constexpr auto kern = gemm_pp128_kernel<EPK, A3>;
    if (int rc = set_dyn_lds<kern>(smem)) return rc;
    const long nbm = (g.M + 255) / 256, nbn = g.N / 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(512), smem, st, g);
    return (int)hipGetLastError();
}
template <int EPK>
static int launch_pp4w32(const GemmArgs& g, hipStream_t st) {
    constexpr int smem = 163840;
    constexpr auto kern = gemm_pp4w_kernel<EPK>;
    if (int rc = set_dyn_lds<kern>(smem)) return rc;
    const long nbm = (g.M + 255) / 256, nbn = g.N / 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(256), smem, st, g);
    return (int)hipGetLastError();
}
template <int WM, int WN, int TM, int EPK, int NS = 4>
static int launch_pp_cfg(const GemmArgs& g, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * 64;
    constexpr int smem = NS * (BM + BN) * 64;
    constexpr auto kern = gemm_pp_kernel<WM, WN, TM, EPK, NS>;
    if (int rc = set_dyn_lds<kern>(smem)) return rc;
    const long nbm = (g.M + BM - 1) / BM, nbn = g.N / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(512), smem, st, g);
    return (int)hipGetLastError();
}
#endif

// Shapes / epilogues the ping-pong kernels take (f16, LINEAR mode only): 256 x 256 tiles of full 128-byte K rows.  The product kernels
// (gemm_pp128p_kernel, persistent, and its one-tile-per-workgroup form gemm_pp128m16_kernel for K < 192: 8 waves each; the 4-wave
// gemm_pp4w16_kernel is an --experiments build only) and the latency-regime kernels of gemm.hip issue v_mfma_f32_16x16x32_f16 over the
// same K grouping and share the epilogue arithmetic, so a GEMM's result does not depend on which of them its size selects
// (tests/test_hip_gemm_pp.py compares them bit for bit, every epilogue flavour).
bool gemm_pp_eligible(const GemmArgs& g) {
    if (g.relu_in || g.add) return false;
    if ((g.N % 256) || (g.K & 63) || g.K < 64 || g.M < 256) return false;
    if ((g.lda & 7) || (g.ldw & 7)) return false;
    if (g.ln_mr) {          // LN-fold consumer: QKV / GELU flavours
        if (!g.ln_c || !g.bias) return false;
        if (!(g.epi == EPI_QKV || (g.epi == EPI_STORE && g.act == ACT_GELU && !g.uv.wu))) return false;
    }
    if (g.x16 && (g.epi != EPI_RESID || (!g.ln_part && g.xres))) return false;                      // (fp16 stream, xres == nullptr: the statistics are optional)
    if (g.epi == EPI_RESID && !g.xres && !g.x16) return false;
    if (g.epi == EPI_RESID && (size_t)g.M * (size_t)g.ldc * 4 >= 0xffffff00ull) return false;       // 32-bit buffer offsets in the RESID epilogue
    switch (g.epi) {
    case EPI_STORE: return (g.ldc & 7) == 0 && (!g.uv.wu || g.bias);
    case EPI_RESID: return g.bias && g.gamma && (g.ldc & 3) == 0;
    case EPI_QKV: return g.v_rowmajor && g.D % 256 == 0 && g.Ntok >= 128 && g.N == 3 * g.D && g.bias;
    case EPI_CONVT: return g.Cout % 64 == 0 && g.pixW >= 8 && !g.uv.wu;
    default: return false;
    }
}

static int epilogue_kind(const GemmArgs& g) {
    switch (g.epi) {
    case EPI_RESID: return g.xres ? EPK_RESID : EPK_RESID16;
    case EPI_QKV: return g.ln_mr ? EPK_QKV_LN : EPK_QKV;
    case EPI_CONVT: return EPK_CONVT;
    default: break;
    }
    if (g.uv.wu) return EPK_UV;
    if (g.act == ACT_GELU) return g.ln_mr ? EPK_GELU_LN : EPK_GELU;
    if (g.act == ACT_RELU) return EPK_RELU;
    return EPK_STORE;
}

// PP_KERN: 2 = gemm_pp128p_kernel (persistent; what -1, the default, selects; K < 192 or outputs beyond 4 GiB fall back to 0),
// 0 = gemm_pp128m16_kernel (one tile per workgroup); -DMOGE_EXPERIMENTS builds: 1 = gemm_pp4w16_kernel (4 waves, 128 x 128 per wave)
template <int EPK>
static int launch_pp_any(const GemmArgs& g, hipStream_t st) {
#ifdef MOGE_EXPERIMENTS
    switch (moge_tune_get("PP_EXP", 0)) {              // tools/kbench A-B only
    case 1: return launch_pp128_cfg<EPK, 1>(g, st);
    case 2: return launch_pp128_cfg<EPK, 2>(g, st);
    case 3: return launch_pp128_cfg<EPK, 0>(g, st);
    case 4: return launch_pp4w32<EPK>(g, st);
    case 5: return launch_pp_cfg<2, 4, 4, EPK>(g, st);
    case 6: return launch_pp_cfg<4, 2, 2, EPK, 3>(g, st);
    default: break;
    }
#endif
    int kern = moge_tune_get("PP_KERN", -1);
    if (kern < 0) kern = 2;
    // The product's persistent kernel is MODE 3 (round 4): the DMA pieces of a K-tile split 4 + 4 over the wave's two read segments (A(t+2) whole
    // in the first, W(t+2) in the second).  Same MFMAs in the same order as every other schedule: bit-identical results.
#ifdef MOGE_EXPERIMENTS
    // MODE 7 (round 6, PP_WX; --experiments builds): the W stream continuous across the tile boundary; needs an even number of K-tiles >= 4.  Bit-identical; kbench,
    // sustained 1500-launch samples: qkv / fc1 / proj +1.2-1.4 %, fc2 level - and 247.1 -> 246.8 img/s in the production two-stream step (EXPERIMENTS R6.3d)
    if (kern == 2 && pp_persistent_ok(g) && moge_tune_get("PP_WX", PP_WX_DEFAULT) != 0 && ((g.K >> 6) & 1) == 0 && g.K >= 256) return launch_pp128p<EPK, 64, 7>(g, st);
#endif
    if (kern == 2 && pp_persistent_ok(g)) return launch_pp128p<EPK, 64, 3>(g, st);
#ifdef MOGE_EXPERIMENTS
    // the other DMA schedules and loop shapes that were measured (tools/kbench A-B builds only; profiles/r04n ... r04r, r03r, r03u):
    //   5: MODE 0, rounds 2-3's 2 + 6 schedule (A_hi(t+2) | W(t+2), A_lo(t+3)): 3-5 % slower at K = 1024, 2-3 % at K = 4096
    //   6: MODE 4, 3 + 5: between the two        7: MODE 5, 4 + 4 with the DMA issued before the fragment reads: 0-1.3 % slower
    //   8: MODE 6, 6 + 2 (half of W(t+1) requested a read segment later): equal at K = 1024, slower than MODE 0 at K = 4096
    //   3: MODE 1, merged 64-MFMA segments: 10-17 % slower        4: MODE 2, W(t+2) requested inside the MFMA segment: 0.5-3 % slower
    if (kern == 5 && pp_persistent_ok(g)) return launch_pp128p<EPK, 64, 0>(g, st);
    if (kern == 6 && pp_persistent_ok(g)) return launch_pp128p<EPK, 64, 4>(g, st);
    if (kern == 7 && pp_persistent_ok(g)) return launch_pp128p<EPK, 64, 5>(g, st);
    if (kern == 8 && pp_persistent_ok(g)) return launch_pp128p<EPK, 64, 6>(g, st);
    if (kern == 3 && pp_persistent_ok(g)) return launch_pp128p<EPK, 64, 1>(g, st);
    if (kern == 4 && pp_persistent_ok(g)) return launch_pp128p<EPK, 64, 2>(g, st);
    if (kern == 1) return launch_pp4w16<EPK>(g, st);
#endif
    return launch_pp128m16<EPK>(g, st);
}

int launch_gemm_pp(const GemmArgs& g0, hipStream_t st) {
    GemmArgs g = g0;
    g.dbg = moge_tune_get("PP_DBG", 0);
    g.abl = moge_tune_get("PP_ABL", 0);
    g.stagger = moge_tune_get("PP_STAGGER", 0);
    g.stagger_clk = moge_tune_get("PP_STAGGER_CLK", 56000);
    {
        const int nt = moge_tune_get("NT_STORE", 0);        // bit 0: GELU (MLP hidden), bit 1: QKV, bit 2: plain stores
        g.nt_store = (g.epi == EPI_QKV) ? (nt >> 1) & 1 : (g.act == ACT_GELU ? nt & 1 : (nt >> 2) & 1);
    }
    switch (epilogue_kind(g)) {
    case EPK_RESID: return launch_pp_any<EPK_RESID>(g, st);
    case EPK_RESID16: return launch_pp_any<EPK_RESID16>(g, st);
    case EPK_QKV: return launch_pp_any<EPK_QKV>(g, st);
    case EPK_CONVT: return launch_pp_any<EPK_CONVT>(g, st);
    case EPK_UV: return launch_pp_any<EPK_UV>(g, st);
    case EPK_GELU: return launch_pp_any<EPK_GELU>(g, st);
    case EPK_GELU_LN: return launch_pp_any<EPK_GELU_LN>(g, st);
    case EPK_QKV_LN: return launch_pp_any<EPK_QKV_LN>(g, st);
    case EPK_RELU: return launch_pp_any<EPK_RELU>(g, st);
    default: return launch_pp_any<EPK_STORE>(g, st);
    }
}
