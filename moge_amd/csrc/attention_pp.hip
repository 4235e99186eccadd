// Flash attention, fp16 throughput path (attention.py:70-81: softmax(q k^T / 8) v, head_dim 64, no mask) for gfx950.
// The fp32 parity mode stays on attention.hip.
//
// Inputs from the QKV GEMM epilogue (EPI_QKV, v_rowmajor): q, k, v as (B, nh, N, 64) fp16, q pre-multiplied by
// log2(e)/8.  Output (B, N, nh*64) fp16.
//
// What differs from attention.hip (measured there: 52 % of wave cycles issuing VALU, MFMA pipe 30 % busy):
//   * The softmax works against a DEFERRED running max m: the accumulator of the S^T = K Q^T MFMA chain is initialised
//     with -m (a 16-register splat that only changes when m does), so the MFMA result is already s - m and the
//     per-score VALU work is ONE v_exp_f32, ONE v_add (row sum) and half a convert: no max tree, no subtract, no
//     accumulator zeroing, no O rescale.  m is raised (exactly, with the usual alpha rescale of O and l) only on the
//     first tile and when some lane's tile sum reaches 2^14, i.e. before any P could overflow fp16; P <= 2^14 relative to
//     a stale m is as accurate in fp16 as P <= 1 (same 11-bit significand), O and l are fp32.
//   * K and V tiles arrive by LDS-DMA (global_load_lds_dwordx4) into a 3-stage ring: one raw s_barrier per 64-key tile,
//     counted vmcnt, the next two tiles in flight during the MFMAs.
//   * V stays row-major (key, d): the PV MFMA's A-operand (V^T rows) is read with ds_read_b64_tr_b16 (hardware
//     transpose), so the QKV epilogue writes V like K in full 128-byte rows instead of 2-byte transposed scatters.
// (A barrier-alternated variant - 8 waves, one group in a 16-MFMA segment while the other runs the softmax / LDS segment, as
// gemm_pp.hip does - was measured at 657-727 TF/s against 885 TF/s for this free-running 3-waves-per-SIMD kernel: the VALU
// issue of the softmax segment is starved by the partner's back-to-back MFMAs, so the extra occupancy wins here.)
// Both products are issued "swapped" (S^T = K Q^T, O^T = V^T P^T) so the query is the lane index and the softmax state is
// per lane; K rows enter the first MFMA in a bit-2/3-swapped order so that its accumulator registers are directly the
// second MFMA's B-operand (see attention.hip).
#include "common.h"

#define AP_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define AP_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 tr_pair(const char* p0, const char* p1) {
    // two transposed 4-key reads -> one 8-key (16-byte) MFMA operand
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(u32x4, v);
}

__device__ __forceinline__ u32x4 pack8(const f32x16& p, int s) {
    f16x8 h;
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = (f16)p[8 * s + i];
    return __builtin_bit_cast(u32x4, h);
}

constexpr int AP_STAGE = 16384;          // K tile 8 KiB + V tile 8 KiB
constexpr float AP_PSUM_LIMIT = 16384.f;

#ifdef MOGE_EXPERIMENTS
#include "../../tools/experiments/attention_pp_exp.inc"     // the v_mfma_f32_32x32x16_f16 form of this kernel (tools/kbench A-B builds only)
#endif

// ------------------------------------------------------------------------------------------------------------------------
// The same kernel on v_mfma_f32_16x16x32_f16 (tools/mfma_power: the cheaper instruction per FLOP on this power-limited chip).
// A wave still owns 32 queries, now as two 16-query column blocks; lane = (query l15, key/d group g4 = lane >> 4):
//   S^T tile (kb, qb) = K[16 keys] Q[16 queries]^T over d in two K-steps of 32;  D: lane holds keys 4*g4 + r of key block kb.
//   Key block kb, row rho = 4*g + r  <->  key 32*(kb >> 1) + 8*g + 4*(kb & 1) + r  (a row permutation applied when the K fragment is
//   read), so that tiles (2s, 2s+1) of a lane are exactly keys 32s + 8*g4 + 0..7: converted to fp16 they ARE the B operand of the
//   O^T += V^T P^T step s (16 queries x 32 keys, lane = query, 8 keys per lane) - no cross-lane movement, as in the 32x32 kernel.
//   V^T operand (16 d x 32 keys): two ds_read_b64_tr_b16 per lane (keys 8*g4 + 0..3 / + 4..7 of the step, the group's 16 d's).
// LDS swizzles (rows of 128 B, 16-byte chunks): K: chunk ^ (((row >> 1) & 1) | (((row >> 3) & 3) << 1));  V: chunk ^ ((((row >> 1) & 1) << 1) |
// (((row >> 3) & 1) << 2)) - both reads conflict-free for these lane patterns.  Softmax state is per (lane, query block); the four lanes
// of a query combine their partial sums / maxima with two shuffles where the exact path needs them.
// ------------------------------------------------------------------------------------------------------------------------
// VAR (bit field; the product instantiates ONE value, AP_VAR_DEFAULT; the others exist in -DMOGE_EXPERIMENTS builds for tools/kbench):
//   1  ablation: the hot loop's exp2 replaced by one multiply (what the transcendental costs; wrong results)
//   2  ablation: no overflow guard at all (what the guard's serialisation costs; wrong on spiked keys)
//   4  K Q^T issue order: all eight (key block, query block) chains take their first K-step, then all eight their second - no MFMA is issued
//      right behind the one it depends on
//   8  overflow guard on the RAW scores (max over the lane's 32 values of s - m, <= 15 -> every P <= 2^15 fits fp16) instead of on the row
//      sums: the decision is known right after the K Q^T chain, so the exponentials are free to interleave with the P V MFMAs (with the
//      sum-based guard the branch sits between the last exponential and the first P V MFMA and serialises the two)
constexpr int AP_VAR_DEFAULT = 0;
constexpr int AP_X_DEFAULT = 0;         // attn_pp16x_kernel (round 5, --experiments builds only): measured 13 % slower than attn_pp16mq<4>
constexpr int AP_KERN_DEFAULT = 3;      // 0: attn_pp16_kernel; 1 / 2 / 3: attn_pp16mq_kernel<2> / <4> / chosen by grid size (row sums on the matrix pipe)
constexpr float AP_SCORE_LIMIT = 15.f;
constexpr float AP_ROWSUM_LIMIT = 49152.f;      // attn_pp16m: FULL row sum of a tile (64 keys) below this -> every P < 49152 < 65504 (fp16 max)
template <int VAR>
__global__ __launch_bounds__(256, 3) void attn_pp16_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v,
                                                           f16* __restrict__ out, int Ntok, int nh) {
    constexpr int NW = 4, NPW = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 3 * AP_STAGE

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / nh, head = bh - b * nh;
    const int q0 = blockIdx.x * (NW * 32) + wave * 32;

    // ---- Q fragments (B operand of S^T): lane = query, d = 32*ks + 8*g4 .. + 7 ----------------------------------------------
    u32x4 qf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
        const int qrow = q0 + qb * 16 + l15;
        const f16* qp = q + ((size_t)bh * Ntok + (qrow < Ntok ? qrow : Ntok - 1)) * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) qf[qb][ks] = *reinterpret_cast<const u32x4*>(qp + 32 * ks + 8 * g4);
    }

    // ---- DMA sources (as attn_pp_kernel; the swizzles differ) ----------------------------------------------------------------
    const char* kbase = reinterpret_cast<const char*>(k + (size_t)bh * Ntok * 64);
    const char* vbase = reinterpret_cast<const char*>(v + (size_t)bh * Ntok * 64);
    const int prow = lane >> 3, pch = lane & 7;
    int drow[NPW];
    unsigned doff[NPW];
#pragma unroll
    for (int i = 0; i < NPW; i++) {
        const int p = wave + NW * i;                     // 0..15: K pieces 0-7, V pieces 8-15
        const int row = (p & 7) * 8 + prow;
        drow[i] = row;
        const int ksw = ((row >> 1) & 1) | (((row >> 3) & 3) << 1);
        const int vsw = (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2);
        doff[i] = (unsigned)(row * 128 + ((pch ^ (p >= 8 ? vsw : ksw)) << 4));
    }
    const int ntiles = (Ntok + 63) >> 6;
    auto issue = [&](int t) {
        char* st = smem + (t % 3) * AP_STAGE;
        const char* kt = uniform_ptr(kbase + (size_t)t * 8192);
        const char* vt = uniform_ptr(vbase + (size_t)t * 8192);
        if (t < ntiles - 1) {
#pragma unroll
            for (int i = 0; i < NPW; i++) {
                const int p = wave + NW * i;
                __builtin_amdgcn_global_load_lds(AP_GPTR((p >= 8 ? vt : kt) + doff[i]), AP_LPTR(st + p * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NPW; i++) {
                const int p = wave + NW * i;
                const int over = t * 64 + drow[i] - (Ntok - 1);
                const unsigned off = doff[i] - (over > 0 ? (unsigned)over * 128u : 0u);
                __builtin_amdgcn_global_load_lds(AP_GPTR((p >= 8 ? vt : kt) + off), AP_LPTR(st + p * 1024), 16, 0, 0);
            }
        }
    };

    // ---- LDS read addresses (bytes inside a stage) -----------------------------------------------------------------------
    // K fragment (kb, ks): row = 32*(kb >> 1) + 4*(kb & 1) + 8*(l15 >> 2) + (l15 & 3) [immediate part: (32*(kb>>1) + 4*(kb&1)) * 128], chunk (4*ks + g4) ^ ksw
    const int kswl = ((l15 >> 1) & 1) | ((l15 >> 2) << 1);
    int kaddr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) kaddr[ks] = (8 * (l15 >> 2) + (l15 & 3)) * 128 + (((4 * ks + g4) ^ kswl) << 4);
    // V^T fragment (s, db), half h: supplier lane li of group g4: key row 32*s + 8*g4 + 4*h + (li >> 2), bytes 32*db + 8*(li & 3) .. + 7
    const int vswl = (((l15 >> 3) & 1) << 1) | ((g4 & 1) << 2);
    int vaddr[4];
#pragma unroll
    for (int db = 0; db < 4; db++)
        vaddr[db] = 8192 + (8 * g4 + (l15 >> 2)) * 128 + ((((2 * db) ^ vswl) | ((l15 & 3) >> 1)) << 4) + (l15 & 1) * 8;

    f32x4 o[4][2];                      // [d block][query block]
    f32x4 negs[2];                      // -m splat per query block (C operand of the first MFMA of a K Q^T chain)
    float m_run[2], l_run[2];
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
#pragma unroll
        for (int db = 0; db < 4; db++) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        negs[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[qb] = -1e30f; l_run[qb] = 0.f;
    }

    issue(0);
    if (ntiles > 1) issue(1);

    // Control flow: the hot loop contains ONLY the fast path; the first tile, the last tile and a tile whose lane sum trips the overflow
    // guard leave it for the exact body (written once, outside).  With the exact body inside the loop the register allocator kept two
    // copies of O / l / -m / S and paid ~60 register moves per tile on the fast edge.  Every path executes exactly one barrier per tile.
    int stage = 0;
    const char* ka[2];
    const char* va[4];
    f32x4 sc[4][2];                  // [key block][query block]
    float psum[2];
    auto tile_head = [&](int t) {         // tile t landed for every wave; every wave is done with tile t-1; keep two tiles in flight
        if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < ntiles) issue(t + 2);
        const int so = stage * AP_STAGE;
        stage = stage == 2 ? 0 : stage + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) ka[ks] = smem + (kaddr[ks] + so);
#pragma unroll
        for (int db = 0; db < 4; db++) va[db] = smem + (vaddr[db] + so);
    };
    auto pv = [&]() {                // O^T += V^T P^T: two steps of 32 keys; the P^T operand of step s is tiles (2s, 2s+1) of the lane
#pragma unroll
        for (int qb = 0; qb < 2; qb++) l_run[qb] += psum[qb];
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            u32x4 pf[2];
#pragma unroll
            for (int qb = 0; qb < 2; qb++) {
                f16x8 hp;
#pragma unroll
                for (int r = 0; r < 4; r++) { hp[r] = (f16)sc[2 * s2][qb][r]; hp[4 + r] = (f16)sc[2 * s2 + 1][qb][r]; }
                pf[qb] = __builtin_bit_cast(u32x4, hp);
            }
#pragma unroll
            for (int db = 0; db < 4; db++) {
                // ablation 32: the V^T operand of step 0 reused for step 1 (half the transposed LDS reads; wrong results)
                const u32x4 vf = tr_pair(va[db] + (32 * ((VAR & 32) ? 0 : s2)) * 128, va[db] + (32 * ((VAR & 32) ? 0 : s2) + 4) * 128);
#pragma unroll
                for (int qb = 0; qb < 2; qb++) mma16<f16>(o[db][qb], vf, pf[qb]);
            }
        }
    };
    auto exact_body = [&](int t, bool last) {      // raise m to the true running max, rescale O and l, P = exp2(s - m)
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            const int koff = (32 * (kb >> 1) + 4 * (kb & 1)) * 128;
            const u32x4 kf0 = *reinterpret_cast<const u32x4*>(ka[0] + koff), kf1 = *reinterpret_cast<const u32x4*>(ka[1] + koff);
#pragma unroll
            for (int qb = 0; qb < 2; qb++) {
                sc[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
                mma16<f16>(sc[kb][qb], kf0, qf[qb][0]);
                mma16<f16>(sc[kb][qb], kf1, qf[qb][1]);
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; qb++) {
            if (last) {
#pragma unroll
                for (int kb = 0; kb < 4; kb++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int key = t * 64 + 32 * (kb >> 1) + 8 * g4 + 4 * (kb & 1) + r;
                        if (key >= Ntok) sc[kb][qb][r] = -1e30f;
                    }
            }
            float mx = sc[0][qb][0];
#pragma unroll
            for (int kb = 0; kb < 4; kb++)
#pragma unroll
                for (int r = 0; r < 4; r++) mx = fmaxf(mx, sc[kb][qb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[qb], mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
            m_run[qb] = m_new;
            l_run[qb] *= alpha;
#pragma unroll
            for (int db = 0; db < 4; db++)
#pragma unroll
                for (int r = 0; r < 4; r++) o[db][qb][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; r++) negs[qb][r] = -m_new;
            float ps = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; kb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    sc[kb][qb][r] = __builtin_amdgcn_exp2f(sc[kb][qb][r] - m_new);
                    ps += sc[kb][qb][r];
                }
            psum[qb] = ps;
        }
        pv();
    };

    tile_head(0);
    exact_body(0, ntiles == 1);
    int t = 1;
    for (;;) {
        bool hit = false;
        for (; t < ntiles - 1; t++) {                // ---- hot loop: S^T - m = K Q^T + (-m);  P = exp2(.) ----
            tile_head(t);
            if constexpr (VAR & 4) {
                u32x4 kf[4][2];
#pragma unroll
                for (int kb = 0; kb < 4; kb++) {
                    const int koff = (32 * (kb >> 1) + 4 * (kb & 1)) * 128;
                    kf[kb][0] = *reinterpret_cast<const u32x4*>(ka[0] + koff);
                    kf[kb][1] = *reinterpret_cast<const u32x4*>(ka[1] + koff);
                }
#pragma unroll
                for (int kb = 0; kb < 4; kb++)
#pragma unroll
                    for (int qb = 0; qb < 2; qb++)
                        sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[kb][0]), __builtin_bit_cast(f16x8, qf[qb][0]), negs[qb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 4; kb++)
#pragma unroll
                    for (int qb = 0; qb < 2; qb++) mma16<f16>(sc[kb][qb], kf[kb][1], qf[qb][1]);
            } else {
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
                const int koff = (32 * (kb >> 1) + 4 * (kb & 1)) * 128;
                const u32x4 kf0 = *reinterpret_cast<const u32x4*>(ka[0] + koff), kf1 = *reinterpret_cast<const u32x4*>(ka[1] + koff);
#pragma unroll
                for (int qb = 0; qb < 2; qb++) {
                    sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf0), __builtin_bit_cast(f16x8, qf[qb][0]), negs[qb], 0, 0, 0);
                    mma16<f16>(sc[kb][qb], kf1, qf[qb][1]);
                }
            }
            }
            bool trig = false;
            if constexpr ((VAR & 8) != 0) {            // guard on the raw scores, decided before any exponential
                float mx = sc[0][0][0];
#pragma unroll
                for (int kb = 0; kb < 4; kb++)
#pragma unroll
                    for (int qb = 0; qb < 2; qb++)
#pragma unroll
                        for (int r = 0; r < 4; r++) mx = fmaxf(mx, sc[kb][qb][r]);
                trig = !(mx <= AP_SCORE_LIMIT);
                if (__builtin_expect(__any(trig), 0)) { hit = true; break; }
            }
#pragma unroll
            for (int qb = 0; qb < 2; qb++) {
                float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
                for (int kb = 0; kb < 4; kb++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        sc[kb][qb][r] = (VAR & 1) ? sc[kb][qb][r] * 1e-3f : __builtin_amdgcn_exp2f(sc[kb][qb][r]);
                        if (!(VAR & 16) || (kb == 0 && r == 0)) { if (kb & 1) ps1 += sc[kb][qb][r]; else ps0 += sc[kb][qb][r]; }   // ablation 16: no row-sum adds
                    }
                psum[qb] = ps0 + ps1;
                trig |= !(psum[qb] < AP_PSUM_LIMIT);
            }
            if constexpr ((VAR & (2 | 8)) == 0) {
                if (__builtin_expect(__any(trig), 0)) { hit = true; break; }
            }
            pv();
        }
        if (!hit) break;
        exact_body(t, false);                        // tile t again from LDS (its barrier and DMA issue are done)
        t++;
    }
    if (ntiles > 1) {
        tile_head(ntiles - 1);
        exact_body(ntiles - 1, true);
    }
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
        float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 16);
        l_tot += __shfl_xor(l_tot, 32);
        const float inv = 1.f / l_tot;
        const int qrow = q0 + qb * 16 + l15;
        if (qrow < Ntok) {
            f16* op = out + ((size_t)b * Ntok + qrow) * ((size_t)nh * 64) + head * 64;
#pragma unroll
            for (int db = 0; db < 4; db++)
                store4(op + 16 * db + 4 * g4, o[db][qb][0] * inv, o[db][qb][1] * inv, o[db][qb][2] * inv, o[db][qb][3] * inv);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// attn_pp16m_kernel: attn_pp16_kernel with the softmax ROW SUMS taken by the matrix pipe.  rocprof + ablations of attn_pp16_kernel (tools/kbench
// KB_ABL, profiles/r02i_kbench_attn_abl.log): the kernel is ISSUE-bound, not MFMA- or LDS-bound - per 16-cycle MFMA a SIMD has ~3 further issue
// slots, the tile needs ~6 (one v_exp_f32 = several slots, one row-sum add, half a convert, 0.75 LDS reads per MFMA): MFMA pipe 48 % busy.
// Removing the 32 row-sum adds alone was worth +18 %, adds + overflow guard +23 %.  Here:
//   * l += 1^T P^T is two extra MFMAs per query block and tile (A operand = a register of fp16 ones): every lane of a query receives the
//     FULL row sum of the tile (no partial sums, no final cross-lane reduction) - 4 MFMAs replace 32 v_add_f32 + the guard's compares;
//   * the same number is the overflow guard: it is formed from the fp16 P values BEFORE any P V MFMA of the tile is issued; a P that
//     overflowed fp16 makes it +inf, any sum >= 2^14 sends the tile to the exact (re-max) path as before.
// The normaliser is now the sum of the fp16-rounded P (exactly what multiplies V) instead of the fp32 P; the exact path (first / last
// tile, guard trips) sums in fp32 and broadcasts the row sum to the query's four lanes so that l has one meaning everywhere.
// ------------------------------------------------------------------------------------------------------------------------
// (Requesting the tile's V^T operands behind the K Q^T MFMAs, so that their LDS latency passes under the exponentials, needs > 168 registers =
// 2 waves per SIMD: measured 783-800 TF/s against 938-944 for this form - the third wave is worth more than the exposed read latency.)
// ------------------------------------------------------------------------------------------------------------------------
// attn_pp16mq_kernel<QB>: the kernel described above with QB 16-query blocks per wave (QB = 2: 32 queries per wave, 3 waves per SIMD - round 2's attn_pp16m_kernel; QB = 4: 64 queries per wave, 256 per
// workgroup, 2 waves per SIMD).  VERDICT r02 item 3: the kernel is ISSUE-bound, so the experiment amortises everything that is per TILE and
// not per query - the K fragment reads, the V^T transposed reads, the DMA issue, the barrier, the loop overhead - over twice the MFMAs
// (per 64 queries and tile: 212 instructions instead of 256), and gives the scheduler four independent K Q^T -> exp -> P V chains per wave
// to interleave.  Result and the reason it is / is not the default: DESIGN.md 4.3 (profiles/r03*_kbench_attn*.log).
// ------------------------------------------------------------------------------------------------------------------------
// The body for one workgroup: head `bh`, queries from `q_base` (this workgroup covers 64 * QB of them; wave w owns 16 * QB from q_base + 16 * QB * w).
// SK (attn_pp16sk_kernel, below): the body runs the key tiles [tb_in, te_in) only and leaves its unnormalised state (O, l relative to the running
// max m) in *acc instead of storing the output; returns false for a wave without queries.  SK = false: all tiles, output stored (the code of rounds 2-5).
// seed_k (SK, attn_pp16ks_kernel's second wave group): LDS address of the sequence's FIRST K tile (the first group's ring, complete behind the first barrier).  The range then
// starts from the running max the unsplit kernel carries out of tile 0 - the same K Q^T MFMAs, the same reduction - and runs its own first tile in the hot loop's form, so its
// P operands are the unsplit kernel's bit for bit (until a guard trip raises m in one range and not in the other) and the two forms differ by fp32 summation order only.
template <int QB> struct AttnAcc { f32x4 o[4][QB]; float m[QB], l[QB]; };
template <int QB, bool SK = false>
__device__ __forceinline__ bool attn_pp16mq_body(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v, f16* __restrict__ out,
                                                 int Ntok, int nh, int bh, int q_base, char* smem, int tb_in = 0, int te_in = 0, AttnAcc<QB>* acc = nullptr,
                                                 const char* seed_k = nullptr) {
    constexpr int NW = 4, NPW = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & (NW - 1);      // (attn_pp16ks_kernel: eight waves = two groups of four, each on its own ring and key range)
    const int l15 = lane & 15, g4 = lane >> 4;
    const int b = bh / nh, head = bh - b * nh;
    const int q0 = q_base + wave * (16 * QB);
    u32x4 qf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const int qrow = q0 + qb * 16 + l15;
        const f16* qp = q + ((size_t)bh * Ntok + (qrow < Ntok ? qrow : Ntok - 1)) * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) qf[qb][ks] = *reinterpret_cast<const u32x4*>(qp + 32 * ks + 8 * g4);
    }
    const char* kbase = reinterpret_cast<const char*>(k + (size_t)bh * Ntok * 64);
    const char* vbase = reinterpret_cast<const char*>(v + (size_t)bh * Ntok * 64);
    const int prow = lane >> 3, pch = lane & 7;
    // DMA source offsets of this lane's pieces: doff for every tile but the last, doff_last for the last one (rows past Ntok - 1 re-read row
    // Ntok - 1; they are masked in exact_block).  Both are loop constants: computing the clamp at issue time put 4 VALU per piece into every
    // iteration of an issue-bound loop (the compiler merges the two sides of `t < ntiles - 1` and selects between the addresses).
    unsigned doff[NPW], doff_last[NPW];
    const int ntiles = (Ntok + 63) >> 6;
    const int tb = SK ? tb_in : 0, te = SK ? te_in : ntiles;          // this call's key tiles
#pragma unroll
    for (int i = 0; i < NPW; i++) {
        const int p = wave + NW * i;
        const int row = (p & 7) * 8 + prow;
        const int ksw = ((row >> 1) & 1) | (((row >> 3) & 3) << 1);
        const int vsw = (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2);
        doff[i] = (unsigned)(row * 128 + ((pch ^ (p >= 8 ? vsw : ksw)) << 4));
        const int over = (ntiles - 1) * 64 + row - (Ntok - 1);
        doff_last[i] = doff[i] - (over > 0 ? (unsigned)over * 128u : 0u);
    }
    auto issue = [&](int t) {
        char* st = smem + ((SK ? t - tb : t) % 3) * AP_STAGE;
        const char* kt = uniform_ptr(kbase + (size_t)t * 8192);
        const char* vt = uniform_ptr(vbase + (size_t)t * 8192);
        const bool last = t >= ntiles - 1;
#pragma unroll
        for (int i = 0; i < NPW; i++) {
            const int p = wave + NW * i;
            __builtin_amdgcn_global_load_lds(AP_GPTR((p >= 8 ? vt : kt) + (last ? doff_last[i] : doff[i])), AP_LPTR(st + p * 1024), 16, 0, 0);
        }
    };
    const int kswl = ((l15 >> 1) & 1) | ((l15 >> 2) << 1);
    int kaddr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) kaddr[ks] = (8 * (l15 >> 2) + (l15 & 3)) * 128 + (((4 * ks + g4) ^ kswl) << 4);
    const int vswl = (((l15 >> 3) & 1) << 1) | ((g4 & 1) << 2);
    int vaddr[4];
#pragma unroll
    for (int db = 0; db < 4; db++)
        vaddr[db] = 8192 + (8 * g4 + (l15 >> 2)) * 128 + ((((2 * db) ^ vswl) | ((l15 & 3) >> 1)) << 4) + (l15 & 1) * 8;

    f32x4 o[4][QB];
    f32x4 negs[QB];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
#pragma unroll
        for (int db = 0; db < 4; db++) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        negs[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[qb] = -1e30f; l_run[qb] = 0.f;
    }
    issue(tb);
    if (te - tb > 1) issue(tb + 1);

    int stage = 0;
    const char* ka[2];
    const char* va[4];
    f32x4 sc[4][QB];
    float psum[QB];
    auto tile_head = [&](int t, bool with_issue = true) {
        if (t + 1 < te) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (with_issue && t + 2 < te) issue(t + 2);
        const int so = stage * AP_STAGE;
        stage = stage == 2 ? 0 : stage + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) ka[ks] = smem + (kaddr[ks] + so);
#pragma unroll
        for (int db = 0; db < 4; db++) va[db] = smem + (vaddr[db] + so);
    };
    auto pack = [&](int qb, u32x4 (&pf)[2]) {          // P^T operands of the two 32-key steps of query block qb
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            f16x8 hp;
#pragma unroll
            for (int r = 0; r < 4; r++) { hp[r] = (f16)sc[2 * s2][qb][r]; hp[4 + r] = (f16)sc[2 * s2 + 1][qb][r]; }
            pf[s2] = __builtin_bit_cast(u32x4, hp);
        }
    };
    auto pv_block = [&](int qb, const u32x4 (&pf)[2]) {
        l_run[qb] += psum[qb];
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int db = 0; db < 4; db++) {
                const u32x4 vf = tr_pair(va[db] + (32 * s2) * 128, va[db] + (32 * s2 + 4) * 128);
                mma16<f16>(o[db][qb], vf, pf[s2]);
            }
    };
    auto exact_block = [&](int t, bool last, int qb) {      // raise m to the true running max, rescale O and l, P = exp2(s - m), P V: one query block
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            const int koff = (32 * (kb >> 1) + 4 * (kb & 1)) * 128;
            const u32x4 kf0 = *reinterpret_cast<const u32x4*>(ka[0] + koff), kf1 = *reinterpret_cast<const u32x4*>(ka[1] + koff);
            sc[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16<f16>(sc[kb][qb], kf0, qf[qb][0]);
            mma16<f16>(sc[kb][qb], kf1, qf[qb][1]);
        }
        if (last) {
#pragma unroll
            for (int kb = 0; kb < 4; kb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int key = t * 64 + 32 * (kb >> 1) + 8 * g4 + 4 * (kb & 1) + r;
                    if (key >= Ntok) sc[kb][qb][r] = -1e30f;
                }
        }
        float mx = sc[0][qb][0];
#pragma unroll
        for (int kb = 0; kb < 4; kb++)
#pragma unroll
            for (int r = 0; r < 4; r++) mx = fmaxf(mx, sc[kb][qb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run[qb], mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        m_run[qb] = m_new;
        l_run[qb] *= alpha;
#pragma unroll
        for (int db = 0; db < 4; db++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[db][qb][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; r++) negs[qb][r] = -m_new;
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; kb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                sc[kb][qb][r] = __builtin_amdgcn_exp2f(sc[kb][qb][r] - m_new);
                ps += sc[kb][qb][r];
            }
        ps += __shfl_xor(ps, 16);                          // full row sum in all four lanes of the query (l is a full sum in this kernel)
        ps += __shfl_xor(ps, 32);
        psum[qb] = ps;
        u32x4 pf[2];
        pack(qb, pf);
        pv_block(qb, pf);
    };
    auto qk_block = [&](int qb, const u32x4 (&kf)[4][2]) {
#pragma unroll
        for (int kb = 0; kb < 4; kb++)
            sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[kb][0]), __builtin_bit_cast(f16x8, qf[qb][0]), negs[qb], 0, 0, 0);
#pragma unroll
        for (int kb = 0; kb < 4; kb++) mma16<f16>(sc[kb][qb], kf[kb][1], qf[qb][1]);
    };
    u32x4 ones;                                             // fp16 1.0 x 8: the A operand of the row-sum MFMA
#pragma unroll
    for (int i = 0; i < 4; i++) ones[i] = 0x3c003c00u;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // A wave whose queries all lie beyond the sequence (the last workgroup of a head: N = 3601 leaves 17 queries for it, i.e. one wave of four)
    // only keeps the workgroup's DMA ring and barriers going - exactly one barrier per tile, like every other path - and leaves: the wave
    // with the real queries has its SIMD to itself and the workgroup frees its slot sooner (its three idle waves used to redo a clamped
    // copy of query N - 1 at full cost).
    if (__builtin_amdgcn_readfirstlane(q0) >= Ntok) {
        for (int tt = tb; tt < te; tt++) tile_head(tt);
        return false;
    }
    bool seeded = false;
    if constexpr (SK) seeded = seed_k != nullptr && tb < ntiles - 1;      // (a range that is only the masked last tile keeps the exact path, from the seeded max)
    tile_head(tb, !seeded);
    if constexpr (SK) if (seed_k) {
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
            float mx = -1e30f;
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
                const int koff = (32 * (kb >> 1) + 4 * (kb & 1)) * 128;
                const u32x4 kf0 = *reinterpret_cast<const u32x4*>(seed_k + kaddr[0] + koff), kf1 = *reinterpret_cast<const u32x4*>(seed_k + kaddr[1] + koff);
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
                mma16<f16>(s0, kf0, qf[qb][0]);
                mma16<f16>(s0, kf1, qf[qb][1]);
#pragma unroll
                for (int r = 0; r < 4; r++) mx = fmaxf(mx, s0[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            m_run[qb] = mx;
#pragma unroll
            for (int r = 0; r < 4; r++) negs[qb][r] = -mx;
        }
    }
    int t = tb + 1;
    if (seeded) t = tb;                              // the first tile in the hot loop's form, against the seeded max (its barrier is done)
    else {
#pragma unroll
        for (int qb = 0; qb < QB; qb++) exact_block(tb, tb == ntiles - 1, qb);
    }
    const int hot_end = SK ? (te < ntiles - 1 ? te : ntiles - 1) : ntiles - 1;      // (the sequence's last tile is masked: exact path)
    u32x4 pf[QB][2];                             // [query block][32-key step]: the tile's P^T operands
    f32x4 lt[QB];                                // this tile's row sums (every register / lane of a query holds the same number)
    for (;;) {
        bool hit = false;
        for (; t < hot_end; t++) {                   // ---- hot loop ----
            if (!(seeded && t == tb)) tile_head(t, false);
            u32x4 kf[4][2];
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
                const int koff = (32 * (kb >> 1) + 4 * (kb & 1)) * 128;
                kf[kb][0] = *reinterpret_cast<const u32x4*>(ka[0] + koff);
                kf[kb][1] = *reinterpret_cast<const u32x4*>(ka[1] + koff);
            }
            // tile t + 2's DMA requests BEHIND the K fragment reads: in front of them the four pieces' issue time sat between the barrier and the first MFMA
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < te) issue(t + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qb = 0; qb < QB; qb++) qk_block(qb, kf);
#pragma unroll
            for (int qb = 0; qb < QB; qb++) {
#pragma unroll
                for (int kb = 0; kb < 4; kb++)
#pragma unroll
                    for (int r = 0; r < 4; r++) sc[kb][qb][r] = __builtin_amdgcn_exp2f(sc[kb][qb][r]);
                pack(qb, pf[qb]);
                lt[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ones), __builtin_bit_cast(f16x8, pf[qb][0]), zero4, 0, 0, 0);
                mma16<f16>(lt[qb], ones, pf[qb][1]);
            }
            bool trig = false;
#pragma unroll
            for (int qb = 0; qb < QB; qb++) trig |= !(lt[qb][0] < AP_ROWSUM_LIMIT);
            if (__builtin_expect(__any(trig), 0)) { hit = true; break; }
#pragma unroll
            for (int qb = 0; qb < QB; qb++) l_run[qb] += lt[qb][0];
            {
                // V^T operand reads AHEAD groups in front of the MFMAs that use them (group = one 16-row block of d x one 32-key step: 2 transposed
                // reads, QB MFMAs; QB = 4: 20 registers in flight - what the compiler's own order held - 256 VGPRs and one spill outside the loop;
                // QB = 2: 161 as before).  Left alone the compiler reads a group, waits lgkmcnt(0), issues its QB MFMAs, four times over (it never
                // counts lgkmcnt here): LDS latency in front of every fourth MFMA.  Same MFMAs in the same order (bit-identical).  kbench, old / new
                // library alternating on one box, 30-100 launches per sample (profiles/r04z*_kbench_attn_ab.log): batch 32 1.590 -> 1.556-1.568 ms
                // and 1.602-1.626 -> 1.561-1.584, batch 16 0.806 -> 0.794, QB = 2 at one / two images 0.067 -> 0.066 / 0.125 -> 0.122.
                // (Tried with it, r04x / r04y: s_setprio 1 around the P V or the K Q^T segment, everywhere but the P V segment, or static by
                //  workgroup - +2 % in 10-launch samples at boost clock, nothing at the clock the chip holds under sustained load.)
                constexpr int AHEAD = QB == 4 ? 5 : 4;
                u32x4 vf[8];
                auto ld = [&](int gi) { vf[gi] = tr_pair(va[gi & 3] + (32 * (gi >> 2)) * 128, va[gi & 3] + (32 * (gi >> 2) + 4) * 128); };
#pragma unroll
                for (int gi = 0; gi < AHEAD; gi++) ld(gi);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int gi = 0; gi < 8; gi++) {
                    if (gi + AHEAD < 8) ld(gi + AHEAD);
#pragma unroll
                    for (int qb = 0; qb < QB; qb++) mma16<f16>(o[gi & 3][qb], vf[gi], pf[qb][gi >> 2]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!hit) break;
        // The decision is per 16-QUERY BLOCK, not per wave: a block whose own row sums stayed below the limit takes exactly the fast path's
        // arithmetic (its P^T operands and row sums are still in registers), a block that tripped is redone exactly from LDS (tile t's barrier
        // and DMA issue are done).  A query's result therefore does not depend on which other queries share its wave - i.e. not on QB, and
        // not on which of the two instantiations a batch size selects.
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
            if (__any(!(lt[qb][0] < AP_ROWSUM_LIMIT))) exact_block(t, false, qb);
            else { psum[qb] = lt[qb][0]; pv_block(qb, pf[qb]); }
        }
        t++;
    }
    if (SK ? (te == ntiles && ntiles - 1 > tb) : (ntiles > 1)) {
        tile_head(ntiles - 1);
#pragma unroll
        for (int qb = 0; qb < QB; qb++) exact_block(ntiles - 1, true, qb);
    }
    if constexpr (SK) {
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
#pragma unroll
            for (int db = 0; db < 4; db++) acc->o[db][qb] = o[db][qb];
            acc->m[qb] = m_run[qb]; acc->l[qb] = l_run[qb];
        }
        return true;
    }
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const float inv = 1.f / l_run[qb];               // already the full row sum
        const int qrow = q0 + qb * 16 + l15;
        if (qrow < Ntok) {
            f16* op = out + ((size_t)b * Ntok + qrow) * ((size_t)nh * 64) + head * 64;
#pragma unroll
            for (int db = 0; db < 4; db++)
                store4(op + 16 * db + 4 * g4, o[db][qb][0] * inv, o[db][qb][1] * inv, o[db][qb][2] * inv, o[db][qb][3] * inv);
        }
    }
    return true;
}

template <int QB>
__global__ __launch_bounds__(256, QB > 2 ? 2 : 3) void attn_pp16mq_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v,
                                                            f16* __restrict__ out, int Ntok, int nh, int xcd_remap, int q_start) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 3 * AP_STAGE
    // XCD-aware order: workgroups are dispatched round-robin over the 8 XCDs in linear block order, so with (x = query block, y = head) the 29
    // query blocks of a head land on all 8 L2s and every L2 streams every head's K / V.  Remapped so that a head's query blocks run on ONE XCD
    // (XCD x takes heads x, x + 8, ...): its K / V tiles are fetched into one L2 once.
    int bh = blockIdx.y, qblk = blockIdx.x;
    if (((gridDim.y & 7) == 0) && (xcd_remap & 1)) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int x = lin & 7, s = lin >> 3;
        const int hs = s / (int)gridDim.x;
        bh = x + 8 * hs;
        qblk = s - hs * (int)gridDim.x;
    }
    const int q_base = q_start + qblk * (64 * QB);     // (q_start > 0: the queries attn_pp16x_kernel left over, see launch_attn_pp16)
    if constexpr (QB == 4) if (xcd_remap & 2) {                 // (bit 1: ATTN_TAIL2, default on)
        // The last workgroup of a head with at most 128 queries left (N = 3601: 17 of 256) runs the 32-queries-per-wave body: its one or two
        // waves that hold real queries then compute 2 query blocks each instead of 4 (two of them clamped copies) - the same arithmetic per
        // query (the two instantiations are bit-identical), half the work on the workgroup that decides when its slot is free again.
        if (Ntok - q_base <= 128) { attn_pp16mq_body<2>(q, k, v, out, Ntok, nh, bh, q_base, smem); return; }
    }
    attn_pp16mq_body<QB>(q, k, v, out, Ntok, nh, bh, q_base, smem);
}

// ------------------------------------------------------------------------------------------------------------------------
// attn_pp16ks_kernel<QB> (round 6, batch-1 latency; VERDICT r05 item 4 "split-KV attention for sub-round grids"): the key range of a query block split
// INSIDE the workgroup.  When the 64 * QB-query blocks of a launch are no more than the chip has CUs (one image of N = 3601 tokens, 16 heads: 240
// blocks of 256 queries), the plain grid leaves every SIMD with one or two waves walking all 57 key tiles - a latency chain.  Here a block is an
// EIGHT-wave workgroup: waves 0-3 run attn_pp16mq's body over the key tiles [0, ceil(ntiles / 2)) on ring 0, waves 4-7 the same queries over the rest on ring 1
// (same barriers: the shorter half adds one), so every SIMD holds two waves and the chain is half as long; then the second group leaves its unnormalised state
// (O, l relative to its running max m) in LDS - the rings are dead by then - and the first group combines in a FIXED order, O = O_0 2^(m_0 - m*) + O_1 2^(m_1 - m*),
// l likewise, normalises and stores.  No global workspace, no atomics, no second launch (the stream-K form of tools/experiments/ paid 7 us for those);
// deterministic run to run.  The second group does not take its running max from its own first tile: it reads the sequence's first K tile out of ring 0 and starts from the
// max the unsplit kernel carries out of tile 0 (seed_k in the body), so both groups' P operands are the unsplit kernel's and only the fp32 summation order differs:
// 0.03-0.15 % of the output values differ from the unsplit kernel's, by one fp16 ulp (profiles/r06al_ks_err.log).  That is still not bit-identity, so a single
// image is no longer bit-identical to the same image inside a large batch in the fp16 modes (ATTN_KS = 0 restores that; the fp32 parity mode runs attention.hip
// and is untouched).  Taken by launch_attn_pp16 for grids of at most one workgroup per CU and at least AP_KS_MIN_TILES key tiles.
constexpr int AP_KS_MIN_TILES = 8;
template <int QB>
__global__ __launch_bounds__(512, 1) void attn_pp16ks_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v,
                                                            f16* __restrict__ out, int Ntok, int nh, int xcd_remap, int tmid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 rings of 3 * AP_STAGE; afterwards the exchange buffer (QB * 18 KiB)
    int bh = blockIdx.y, qblk = blockIdx.x;
    if (((gridDim.y & 7) == 0) && (xcd_remap & 1)) {                 // a head's query blocks on ONE XCD (see attn_pp16mq_kernel)
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int x = lin & 7, s = lin >> 3;
        const int hs = s / (int)gridDim.x;
        bh = x + 8 * hs;
        qblk = s - hs * (int)gridDim.x;
    }
    const int q_base = qblk * (64 * QB);
    const int tid = threadIdx.x, lane = tid & 63, t4 = tid & 255;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave8 >> 2, wave = wave8 & 3;
    const int ntiles = (Ntok + 63) >> 6;                             // tmid (host): the first group's tile count, ceil(ntiles / 2) unless a test moves it (1 <= tmid < ntiles)
    AttnAcc<QB> a;
    const bool active = attn_pp16mq_body<QB, true>(q, k, v, out, Ntok, nh, bh, q_base, smem + grp * (3 * AP_STAGE), grp ? tmid : 0, grp ? ntiles : tmid, &a, grp ? smem : nullptr);
    // one barrier per tile in the body: the group with fewer tiles makes up the difference (the second one by one tile when ntiles is odd)
    for (int i = grp ? ntiles - tmid : tmid, n = max(tmid, ntiles - tmid); i < n; i++) __builtin_amdgcn_s_barrier();
    __syncthreads();                                                 // every wave is done with both rings
    f32x4* ex_o = reinterpret_cast<f32x4*>(smem);                    // [db * QB + qb][lane of the group] fragment-linear: conflict-free 16-byte accesses
    f32x2* ex_ml = reinterpret_cast<f32x2*>(smem + (size_t)4 * QB * 256 * 16);
    if (grp && active) {
#pragma unroll
        for (int db = 0; db < 4; db++)
#pragma unroll
            for (int qb = 0; qb < QB; qb++) ex_o[(db * QB + qb) * 256 + t4] = a.o[db][qb];
#pragma unroll
        for (int qb = 0; qb < QB; qb++) ex_ml[qb * 256 + t4] = f32x2{a.m[qb], a.l[qb]};
    }
    __syncthreads();
    if (grp || !active) return;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int b = bh / nh, head = bh - b * nh;
    const int q0 = q_base + wave * (16 * QB);
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const f32x2 ml = ex_ml[qb * 256 + t4];
        const float mstar = fmaxf(a.m[qb], ml[0]);
        const float w0 = __builtin_amdgcn_exp2f(a.m[qb] - mstar), w1 = __builtin_amdgcn_exp2f(ml[0] - mstar);
        const float l = fmaf(w1, ml[1], w0 * a.l[qb]);
        const float inv = 1.f / l;
        const int qrow = q0 + qb * 16 + l15;
#pragma unroll
        for (int db = 0; db < 4; db++) {
            const f32x4 o1 = ex_o[(db * QB + qb) * 256 + t4];
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = fmaf(w1, o1[r], w0 * a.o[db][qb][r]);
            if (qrow < Ntok) {
                f16* op = out + ((size_t)b * Ntok + qrow) * ((size_t)nh * 64) + head * 64;
                store4(op + 16 * db + 4 * g4, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
            }
        }
    }
}

#ifdef MOGE_EXPERIMENTS
#include "../../tools/experiments/attention_pp16sk_exp.inc"    // attn_pp16sk_kernel: attn_pp16mq's body over a stream-K partition of (query block, key tile) units (round 6; correct, 5-8 us SLOWER at one image)
#else
size_t attention_pp_ws_bytes(int, int, int) { return 0; }             // (the stream-K form and its workspace exist in --experiments builds only)
size_t attention_pp_ws_counter_bytes(int, int, int) { return 0; }
#endif

#ifdef MOGE_EXPERIMENTS
#include "../../tools/experiments/attention_pp16s_exp.inc"     // attn_pp16s_kernel: block-pipelined single-stream form (round 6; bit-identical, 11-27 % slower)
#include "../../tools/experiments/attention_pp16x_exp.inc"     // attn_pp16x_kernel: the 8-wave ping-pong form (round 5; bit-identical, 13 % slower: DESIGN.md)
#endif

// (A software-pipelined variant - K Q^T of tile t issued under the softmax of tile t-1, two S buffers, four-stage ring, 2 waves per SIMD -
// was written and measured: correct, 860 TF/s against 920 for the kernel above.  With the schedule left to the compiler the exps still
// cluster (18 in a row between MFMAs) and the second S buffer costs register moves / spills; it needs a hand-placed instruction stream.)

static int launch_attn_pp16(const void* q, const void* k, const void* v, void* out, int B, int nh, int Ntok, hipStream_t st, void* ws, size_t ws_bytes) {
    constexpr int smem = 3 * AP_STAGE;
#ifdef MOGE_EXPERIMENTS
    {
        int QB, nx, gx;
        if (ws && attn_sk_plan(B, nh, Ntok, &QB, &nx, &gx) && ws_bytes >= attention_pp_ws_bytes(B, nh, Ntok)) {
            int* cnt = reinterpret_cast<int*>(ws);
            float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + attention_pp_ws_counter_bytes(B, nh, Ntok));
            if (QB == 4) {
                if (int rc = set_dyn_lds<attn_pp16sk_kernel<4>>(smem)) return rc;
                hipLaunchKernelGGL(attn_pp16sk_kernel<4>, dim3(nx * gx), dim3(256), smem, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, nx, B * nh / nx, part, cnt);
            } else {
                if (int rc = set_dyn_lds<attn_pp16sk_kernel<2>>(smem)) return rc;
                hipLaunchKernelGGL(attn_pp16sk_kernel<2>, dim3(nx * gx), dim3(256), smem, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, nx, B * nh / nx, part, cnt);
            }
            return (int)hipGetLastError();
        }
    }
#else
    (void)ws; (void)ws_bytes;
#endif
    dim3 grid((Ntok + 127) / 128, B * nh);
    // attn_pp16mq_kernel<QB>: row sums on the matrix pipe; QB = 4 (64 queries per wave, 2 waves per SIMD) when the grid still fills the chip
    // several times over, QB = 2 (3 waves per SIMD) otherwise - bit-identical results (see the kernel).  ATTN_KERN: 0 = attn_pp16_kernel,
    // 1 = QB 2 always, 2 = QB 4 always, 3 (default) = by grid size
    const int akern = moge_tune_get("ATTN_KERN", AP_KERN_DEFAULT);
    if (akern >= 1) {
        // QB = 2 while its grid fits ONE round of the chip (3 workgroups per CU): a lone round is a latency chain per workgroup and the 32-query
        // form's is the shorter one (N = 3601, one image: 464 workgroups, 720 vs 680 TF/s; N = 1370 x 4: 760 vs 631); from two images on the
        // 64-query form wins (928 workgroups = 1.2 rounds against 480 = 0.94: 745-797 vs 845-879 TF/s; r03zd_kbench_attn_b1.log)
        const long wgs2 = (long)((Ntok + 127) / 128) * B * nh;
        const bool q4 = akern == 2 || akern == 4 || (akern == 3 && wgs2 > moge_tune_get("ATTN_Q2_MAX_WGS", 3 * pp_device_cus()));
        const int xr = (moge_tune_get("ATTN_XCD", 1) ? 1 : 0) | (moge_tune_get("ATTN_TAIL2", 1) ? 2 : 0);
#ifdef MOGE_EXPERIMENTS
        // attn_pp16x_kernel (ping-pong wave groups, 512 queries per workgroup, ONE workgroup per CU): bit-identical to the two above; taken while
        // its grid still fills the chip at least ATTN_X_MIN_ROUNDS (default 2) times.  ATTN_KERN 4 forces it, ATTN_X = 0 switches it off.
        const int nfull = Ntok / 512, rem = Ntok - nfull * 512;
        const int nblk = nfull + (rem > 256 ? 1 : 0);                 // a block more than half full runs here, a smaller rest on attn_pp16mq
        const bool xk = nblk > 0 && (akern == 4 || (akern == 3 && moge_tune_get("ATTN_X", AP_X_DEFAULT) &&
                                                    (long)nblk * B * nh >= (long)moge_tune_get("ATTN_X_MIN_ROUNDS", 2) * pp_device_cus()));
        if (xk) {
            constexpr int smem_x = 4 * AP_STAGE + 8 * 8192;
            if (moge_tune_get("ATTN_X_TS", 0)) {                // tools only: segment timeline of one workgroup (see the kernel)
                static unsigned long long* dts = nullptr;
                if (!dts && hipMalloc(&dts, 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
                (void)hipMemsetAsync(dts, 0, 16 * sizeof(unsigned long long), st);
                if (int rc = set_dyn_lds<attn_pp16x_kernel<0, 1>>(smem_x)) return rc;
                hipLaunchKernelGGL((attn_pp16x_kernel<0, 1>), dim3(nblk, B * nh), dim3(512), smem_x, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr, dts);
                unsigned long long h[16];
                (void)hipMemcpyAsync(h, dts, sizeof(h), hipMemcpyDeviceToHost, st);
                (void)hipStreamSynchronize(st);
                for (int g2 = 0; g2 < 2; g2++)
                    if (h[g2 * 8 + 4])
                        fprintf(stderr, "[attn_pp16x ts] group %d: %llu hot tiles; per tile (s_memtime ticks): E body %.0f, barrier wait %.0f, M body %.0f, barrier wait %.0f\n", g2, h[g2 * 8 + 4],
                                (double)h[g2 * 8 + 0] / h[g2 * 8 + 4], (double)h[g2 * 8 + 1] / h[g2 * 8 + 4], (double)h[g2 * 8 + 2] / h[g2 * 8 + 4], (double)h[g2 * 8 + 3] / h[g2 * 8 + 4]);
            } else if (moge_tune_get("ATTN_X_PRIO", 0)) {
                if (int rc = set_dyn_lds<attn_pp16x_kernel<1>>(smem_x)) return rc;
                hipLaunchKernelGGL(attn_pp16x_kernel<1>, dim3(nblk, B * nh), dim3(512), smem_x, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr);
            } else {
                if (int rc = set_dyn_lds<attn_pp16x_kernel<0>>(smem_x)) return rc;
                hipLaunchKernelGGL(attn_pp16x_kernel<0>, dim3(nblk, B * nh), dim3(512), smem_x, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr);
            }
            const int covered = nblk * 512;
            if (covered < Ntok) {
                if (int rc = set_dyn_lds<attn_pp16mq_kernel<4>>(smem)) return rc;
                hipLaunchKernelGGL(attn_pp16mq_kernel<4>, dim3((Ntok - covered + 255) / 256, B * nh), dim3(256), smem, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr, covered);
            }
            return (int)hipGetLastError();
        }
#endif
#ifdef MOGE_EXPERIMENTS
        // attn_pp16s_kernel (software-pipelined, one wave per SIMD): the FULL 256-query blocks of every head; the rest of a head (N = 3601: 17 queries) on
        // attn_pp16mq<4>'s tail path.  Bit-identical to attn_pp16mq<2> / <4>.  ATTN_KERN 6 forces it (7: its two-workgroups-per-CU build)
        const int sfull = Ntok / 256;
        if ((akern == 6 || akern == 7) && sfull > 0 && Ntok > 128) {
            constexpr int smem_s = APS_NSTG * AP_STAGE + 4 * 8192;      // ring + the four waves' Q fragments
            if (akern == 6) {
                if (int rc = set_dyn_lds<attn_pp16s_kernel<1>>(smem_s)) return rc;
                hipLaunchKernelGGL(attn_pp16s_kernel<1>, dim3(sfull, B * nh), dim3(256), smem_s, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr);
            } else {
                constexpr int smem_s2 = 5 * AP_STAGE;
                if (int rc = set_dyn_lds<attn_pp16s_kernel<2>>(smem_s2)) return rc;
                hipLaunchKernelGGL(attn_pp16s_kernel<2>, dim3(sfull, B * nh), dim3(256), smem_s2, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr);
            }
            const int covered = sfull * 256;
            if (covered < Ntok) {
                if (int rc = set_dyn_lds<attn_pp16mq_kernel<4>>(smem)) return rc;
                hipLaunchKernelGGL(attn_pp16mq_kernel<4>, dim3((Ntok - covered + 255) / 256, B * nh), dim3(256), smem, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr, covered);
            }
            return (int)hipGetLastError();
        }
#endif
        // attn_pp16ks_kernel: grids of at most ONE 4-wave workgroup per CU run as 8-wave workgroups with the key range split inside the workgroup
        // (two waves per SIMD, half the serial tiles; see the kernel).  32-query blocks while they fit the chip (more CUs busy), else 64-query blocks
        // (N = 3601, 16 heads, one image: 464 > 256 >= 240).  ATTN_KS: 0 off (a single image is then bit-identical to a batch item), 1 by grid size (default), 2 / 4 forced (tests)
        if (const int ks = moge_tune_get("ATTN_KS", 1); ks > 1 || (ks == 1 && akern == 3)) {
            constexpr int smem_ks = 6 * AP_STAGE;
            const int cus = pp_device_cus(), ntiles = (Ntok + 63) >> 6;
            const long w2 = (long)((Ntok + 127) / 128) * B * nh, w4 = (long)((Ntok + 255) / 256) * B * nh;
            const bool fits = ks == 1 && ntiles >= AP_KS_MIN_TILES;
            int tmid = (ntiles + 1) >> 1;
            if (const int f = moge_tune_get("ATTN_KS_MID", 0); f > 0 && f < ntiles) tmid = f;       // tests / tools: any split point
            if ((ks == 2 && ntiles >= 2) || (fits && w2 <= cus)) {
                if (int rc = set_dyn_lds<attn_pp16ks_kernel<2>>(smem_ks)) return rc;
                hipLaunchKernelGGL(attn_pp16ks_kernel<2>, dim3((Ntok + 127) / 128, B * nh), dim3(512), smem_ks, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr, tmid);
                return (int)hipGetLastError();
            }
            if ((ks == 4 && ntiles >= 2) || (fits && w4 <= cus)) {
                if (int rc = set_dyn_lds<attn_pp16ks_kernel<4>>(smem_ks)) return rc;
                hipLaunchKernelGGL(attn_pp16ks_kernel<4>, dim3((Ntok + 255) / 256, B * nh), dim3(512), smem_ks, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr, tmid);
                return (int)hipGetLastError();
            }
        }
        if (q4) {
            if (int rc = set_dyn_lds<attn_pp16mq_kernel<4>>(smem)) return rc;
            hipLaunchKernelGGL(attn_pp16mq_kernel<4>, dim3((Ntok + 255) / 256, B * nh), dim3(256), smem, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr, 0);
        } else {
            if (int rc = set_dyn_lds<attn_pp16mq_kernel<2>>(smem)) return rc;
            hipLaunchKernelGGL(attn_pp16mq_kernel<2>, grid, dim3(256), smem, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh, xr, 0);
        }
        return (int)hipGetLastError();
    }
#define AP_LAUNCH_VAR(V) do { if (int rc = set_dyn_lds<attn_pp16_kernel<V>>(smem)) return rc; \
        hipLaunchKernelGGL(attn_pp16_kernel<V>, grid, dim3(256), smem, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh); \
        return (int)hipGetLastError(); } while (0)
#ifdef MOGE_EXPERIMENTS
    switch (moge_tune_get("ATTN_VAR", AP_VAR_DEFAULT)) {       // tools/kbench A-B only
    case 0: AP_LAUNCH_VAR(0);
    case 1: AP_LAUNCH_VAR(1);
    case 2: AP_LAUNCH_VAR(2);
    case 4: AP_LAUNCH_VAR(4);
    case 6: AP_LAUNCH_VAR(6);
    case 8: AP_LAUNCH_VAR(8);
    case 12: AP_LAUNCH_VAR(12);
    case 16: AP_LAUNCH_VAR(16);
    case 18: AP_LAUNCH_VAR(18);
    case 19: AP_LAUNCH_VAR(19);
    case 32: AP_LAUNCH_VAR(32);
    case 50: AP_LAUNCH_VAR(50);
    case 51: AP_LAUNCH_VAR(51);
    default: return -1;
    }
#else
    AP_LAUNCH_VAR(AP_VAR_DEFAULT);
#endif
#undef AP_LAUNCH_VAR
}

#ifdef MOGE_EXPERIMENTS
template <int NW, int QT>
static int launch_attn_pp_cfg(const void* q, const void* k, const void* v, void* out, int B, int nh, int Ntok, hipStream_t st) {
    constexpr int smem = 3 * AP_STAGE;
    constexpr auto kern = attn_pp_kernel<NW, QT>;
    if (int rc = set_dyn_lds<kern>(smem)) return rc;
    dim3 grid((Ntok + NW * 32 * QT - 1) / (NW * 32 * QT), B * nh);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, Ntok, nh);
    return (int)hipGetLastError();
}
#endif

// q, k, v: (B, nh, Ntok, 64) fp16 (q pre-scaled by log2(e)/8); out: (B, Ntok, nh*64) fp16.  ws (optional): attention_pp_ws_bytes(B, nh, Ntok) bytes whose first
// attention_pp_ws_counter_bytes() are zero - enables the stream-K form on sub-round grids
int launch_attention_pp(const void* q, const void* k, const void* v, void* out, int B, int nh, int Ntok, hipStream_t st, void* ws, size_t ws_bytes) {
    if (Ntok < 1) return -1;
#ifdef MOGE_EXPERIMENTS
    switch (moge_tune_get("ATTN_EXP", 0)) {            // tools/kbench A-B only
    case 1: return launch_attn_pp_cfg<4, 1>(q, k, v, out, B, nh, Ntok, st);
    case 2: return launch_attn_pp_cfg<4, 2>(q, k, v, out, B, nh, Ntok, st);
    case 3: return launch_attn_pp_cfg<8, 1>(q, k, v, out, B, nh, Ntok, st);
    default: break;
    }
#endif
    return launch_attn_pp16(q, k, v, out, B, nh, Ntok, st, ws, ws_bytes);
}
