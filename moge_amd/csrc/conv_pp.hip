// Replicate-padded 3x3 convolution (modules.py:53,59,148-181) as a ping-pong MFMA kernel with an LDS-resident halo tile,
// fp16 throughput path for gfx950.  The fp32 parity mode and unsupported shapes stay on gemm.hip (AMODE_CONV3).
//
//   out[b,y,x,n] = bias[n] + sum_{tap=(dy,dx), c} W[n][tap*Cin + c] * in[b, clamp(y+dy), clamp(x+dx), c]       (NHWC, fp16)
//
// gemm.hip runs this as an implicit GEMM whose A rows are gathered from global memory once PER TAP: nine L2 reads of every
// input pixel, which is what bounds the narrow layers (Cout = 64 / 32x4: measured 380 TF/s).  Here a workgroup owns a
// 16 x 16 pixel tile and BN output channels:
//   * per 64-channel chunk of Cin the (16+2) x (16+2) pixel halo is copied ONCE into LDS by LDS-DMA (source indices
//     clamped = replicate padding for free); the nine taps read shifted fragments of the same LDS image;
//   * only the weight tile of a (chunk, tap) K-step (BN rows x 128 B) is streamed per step, through a 4-slot ring,
//     three steps ahead, with counted vmcnt;
//   * the 8 waves run as two groups half a step apart (one computing 16 or 8 MFMAs while the other reads its fragments),
//     exactly like gemm_pp.hip; the epilogue transposes through LDS and stores 16 bytes per lane in full pixel rows.
// Options: ReLU on the input (applied to the fragments after the LDS read), bias, uv rank-2 term, ReLU, residual add
// (in place allowed), the pixel-shuffle store of the 4-phase "bilinear x2 + 3x3" resampler (EPI_CONVT), and a fused 1x1
// SIDE INPUT: out += W2 . a2 (same pixel, C2 = Cin channels) - the head's `x + in_l(neck_l)` (modules.py:245) runs as extra
// K-steps (centre tap of a2's halo) instead of a separate HBM-bound pass over the level's activations.
// PERSISTENT for the forms that fit one workgroup per CU (BN = 128, or two halo buffers): a workgroup walks a list of tiles and requests the next
// tile's first halo image from the epilogue of the current one (the epilogue stages through the weight ring, so both halo buffers are free);
// the single-image 64-channel form (two workgroups per CU) keeps one tile per workgroup.  See PERSIST in the kernel.
// Tile width TW = 16, or 32 for the 64-channel layers (64 px x 64 ch per wave: 16 fragment reads per 16 MFMAs instead of 12
// per 8 - that configuration is LDS-read bound).
#include "common.h"

#define CP_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
// per-lane 32-bit DMA source offset made opaque at the point of use: keeps  uniform base + zext(offset)  inside the loop, which selects the SADDR form of
// global_load_lds (no 64-bit VALU add per piece; see pp_opaque in gemm_pp.hip)
__device__ __forceinline__ unsigned cp_opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }
#define CP_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

template <int TW> struct Halo {
    static constexpr int W = TW + 2;                     // halo row width in pixels
    static constexpr int PX = 18 * W;                    // 16 + 2 rows
    static constexpr int PIECES = (PX + 7) / 8;          // DMA pieces of 8 pixels x 128 B (TW 16: 41, TW 32: 77)
    static constexpr int BYTES = PIECES * 1024;
    static constexpr int HPW = (PIECES + 7) / 8;         // pieces per wave; surplus pieces repeat the last one
};

__device__ __forceinline__ u32x4 relu8(u32x4 v) {
    // one v_pk_max_f16 per dword (the C++ form compiles to pk_max + compare + select + permute: NaN canonicalisation)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned t = v[i];
        asm("v_pk_max_f16 %0, %0, 0" : "+v"(t));
        v[i] = t;
    }
    return v;
}

template <int N> __device__ __forceinline__ void wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// BN = 128: waves 4 (pixels) x 2 (channels);  BN = 64: waves 8 x 1.  A wave owns TM = TW*16/32/WM tiles of 32 pixels x 64 channels.
// NH = halo buffers (1: Cin == 64 and no side input);  EPI: bit 0 = uv term, bit 1 = pixel-shuffle (EPI_CONVT) store, bit 2 = ReLU on the input
template <int BN, int TW, int NH, int EPI>
__global__ __launch_bounds__(512, (BN == 64 && TW == 16 && NH == 1) ? 4 : 2) void conv_pp_kernel(const GemmArgs g) {
    constexpr bool CONVT = (EPI & 2) != 0, HAS_UV = (EPI & 1) != 0, RELU_IN = (EPI & 4) != 0;
    // EPI bit 3: v_mfma_f32_16x16x32_f16 (as gemm_pp128m16): a 16-pixel block is one row of the 16-wide tile (pixel = lane & 15, channel group
    // lane >> 4), accumulators f32x4[2*TM][4]; the halo / weight LDS layouts, DMA, waits and barriers are unchanged.  TW = 16 only.
    constexpr bool M16 = (EPI & 8) != 0;
    static_assert(!M16 || TW == 16, "16x16x32 path: 16-pixel-wide tiles");
    // EPI bit 4: the fused 1x1 side input of a 64-channel layer WITHOUT a second halo buffer: a 1x1 conv needs only the tile's own pixels, so
    // each lane loads its four B fragments of the side map (2 pixel blocks x 2 K-steps, 64 bytes) straight into registers at the tile head and
    // the side conv is a tenth K-step on them.  Keeps the single-image form's 73 KiB of LDS = TWO workgroups per CU (the two-halo-buffer form
    // of round 2 fits one: 1176 us against ~600 us for the plain conv of the same shape).
    constexpr bool SIDE_REG = (EPI & 16) != 0;
    // EPI bit 5 (with bit 1, BN = 128, Cout = 32): the pixel-shuffled 32-channel map is NOT stored; every high-res pixel's channels are rounded
    // to fp16 (what the store would have kept) and contracted with up to 3 x 4 output-conv rows by ONE small MFMA per group, block and phase:
    // the accumulator layout (lane = pixel, channels 16*jj + 4*g4 + e) is already a B operand whose K index is a permutation of the 32
    // channels; the A operand (g.dot_tab) holds the weight rows in the same permutation, replicated over the four row groups so that every
    // lane group receives the four outputs.  modules.py:231 (+ :245 pre-composed, see model.hip): the 960 x 960 x 32 maps of level 4 were
    // 2 x 1.9 GB written and 4 x 1.9 GB read per step (head_final) for 3 + 1 useful channels.
    constexpr bool DOT = (EPI & 32) != 0;
    static_assert(!DOT || (CONVT && M16 && BN == 128), "fused output conv: the 64 -> 4 x 32 pixel-shuffle resampler");
    // EPI bit 6 (with bits 1 and 3; round 6): CT3 - ConvTranspose2d(k2, s2) and the 3x3 conv behind it (modules.py:160-165) as ONE conv on the LOW-res map.
    // Output pixel (2y + py, 2x + px) sees the low-res pixels (y + py - 1 + tdy, x + px - 1 + tdx), tdy, tdx in {0, 1}: four taps per phase, the weights
    // of (phase, tap) pre-composed at pack time (sum over the 3x3 taps that fall on that low-res pixel of W3 . WT, elementwise.hip compose_ct3).  A workgroup
    // owns ONE phase (Cout = 128; column block = phase) or the two phases of a row parity (Cout = 64, see set_abase), its K loop is 4 taps x Cin / 64 chunks on the same 18 x 18 halo image the 3x3 kernel loads, the
    // pixel-shuffle epilogue stores (B, 2H, 2W, Cout).  16 Cin Cout MACs per output pixel instead of 2 Cin Cout + 9 Cout Cout... (Cin = 2 Cout: 16 vs 22 Cout^2),
    // and the (B, 2H, 2W, Cout) map between the two never exists.  The fused 1x1 side input (heads) reads the HIGH-res map a2 at (2y + py, 2x + px).
    // Replicate padding does not commute with the transposed conv: on the outermost ring of output pixels the composed form equals a REFLECTING pad;
    // launch_ct3_border (below) adds the difference for those pixels afterwards.
    constexpr bool CT3 = (EPI & 64) != 0;
    static_assert(!CT3 || (CONVT && M16 && !DOT && !RELU_IN && !SIDE_REG && NH == 2), "fused ConvTranspose2d + 3x3: pixel-shuffle store, 16x16x32 form, two halo buffers");
    constexpr int NTAP = CT3 ? 4 : 9;
    static_assert(!SIDE_REG || (M16 && BN == 64 && NH == 1), "register side input: single-image 64-channel form");
    constexpr int WN = BN / 64, WM = 8 / WN, TM = TW * 16 / 32 / WM, TN = 2;
    constexpr int HALO_W = Halo<TW>::W, HALO_PX = Halo<TW>::PX, HALO_PIECES = Halo<TW>::PIECES, HALO_BYTES = Halo<TW>::BYTES, HPW = Halo<TW>::HPW;
    constexpr int LOG_TW = TW == 16 ? 4 : 5;
    // The forms that fit ONE workgroup per CU walk a tile list (see below); the single-image 64-channel form fits two per CU, which overlap each
    // other's prologue and epilogue already - persistent it measured 4-6 % slower (its 9-step tiles are short, and the weight ring, which the
    // persistent epilogue stages through, can only be refilled at the tile head): it keeps one tile per workgroup.
    constexpr bool PERSIST = !(BN == 64 && NH == 1);
    constexpr int NWP = BN / 64;                                         // weight pieces (8 rows x 128 B) per wave per K-step
    constexpr int WSLOT = BN * 128;
    constexpr int LDS_W = NH * HALO_BYTES;                                // weight ring behind the two halo buffers
    constexpr int WROWS = TM * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 halos + 4 weight slots

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;

    const int H = g.H, W = g.W, C = g.C;
    const int tx_n = (W + TW - 1) / TW, ty_n = (H + 15) >> 4;
    const int nbn = g.N / BN;
    // PERSISTENT: a workgroup walks a list of tiles.  Each XCD owns a contiguous range of the tile order (channel block fastest, then x, y,
    // image), its workgroups (blockIdx & 7 = XCD) take consecutive tiles of it round by round - neighbouring tiles share halo pixels and the
    // weight tile in that XCD's L2.  Why persistent: an s_memtime timeline (tools/kbench KB_TS, round 2) put 5-6 k clocks of a 64-channel
    // tile's 17-20 k into the PROLOGUE (cold halo DMA from HBM before the first MFMA), 15-18 % for the 128 / 256-channel tiles: here the next
    // tile's first halo image is requested from the epilogue of the current one and lands while it runs.
    int li, cnt, start, wgs_x;
    {
        const int nwg = gridDim.x;
        // (one tile per workgroup: the grid IS the tile count - no division in front of every tile)
        const int ntiles = PERSIST ? (g.M / (H * W)) * ty_n * tx_n * nbn : nwg;
        const int nx = nwg < 8 ? nwg : 8, xcd = blockIdx.x % nx;          // (fewer than 8 workgroups: as many ranges as workgroups)
        const int q = ntiles / nx, r = ntiles % nx;
        cnt = q + (xcd < r ? 1 : 0);
        start = xcd * q + min(xcd, r);
        wgs_x = (nwg - xcd + nx - 1) / nx;
        li = blockIdx.x / nx;
    }
    if (li >= cnt) return;
    int b, y0, x0, n0;                                  // the tile this workgroup is computing / storing
    const int nchunks = NH == 1 ? 1 : (C >> 6);        // NH == 1: Cin == 64, straight-line 9-step K loop
    const int C2 = CT3 ? g.Cout : C;                   // channels of the side map (CT3: the high-res map has Cout channels)
    const int nside = SIDE_REG ? 1 : ((NH > 1 && g.a2) ? (C2 >> 6) : 0);  // fused 1x1 side input: one centre-tap K-step per 64 channels of a2
    const int nkt1 = nchunks * NTAP;
    const int nkt = nkt1 + nside;

    // ---- DMA sources ----------------------------------------------------------------------------------------------
    const int prow = lane >> 3, pch = lane & 7;
    const char* in_b;                   // image base of the tile whose inputs are being requested (conv input / side input: same NHWC shape)
    const char* in2_b;
    const char* w_b;
    const char* w2_b;                   // [N][C] 1x1 weights of the side input
    unsigned hoff[HPW];                 // byte offset of this lane's source chunk for halo piece i (chunk 0 of Cin)
    unsigned hoff2[CT3 ? HPW : 1];      // CT3: the same for the side map (high-res pixel (2 yy + py, 2 xx + px), Cout channels)
    int sb, sy0, sx0, sn0;              // coordinates of that tile
    auto setup = [&](int idx) {
        // (opaque per call: the per-piece halo coordinates below are tile-invariant, and hoisted out of the tile loop they cost ~20 VGPRs that the
        //  128-register forms can only keep in scratch)
        int prow_t = prow, pch_t = pch;
        if constexpr (PERSIST) asm volatile("" : "+v"(prow_t), "+v"(pch_t));
        const int wgt = start + idx;
        const int bn_ = wgt % nbn;
        int t = wgt / nbn;
        const int tx = t % tx_n; t /= tx_n;
        const int ty = t % ty_n;
        sb = t / ty_n; sy0 = ty * 16; sx0 = tx * TW; sn0 = bn_ * BN;
        in_b = reinterpret_cast<const char*>(g.a) + (size_t)sb * H * W * C * 2;
        in2_b = reinterpret_cast<const char*>(g.a2) + (size_t)sb * H * W * C2 * (CT3 ? 8 : 2);
        w_b = reinterpret_cast<const char*>(g.w) + (size_t)sn0 * g.ldw * 2;
        w2_b = reinterpret_cast<const char*>(g.w2) + (CT3 ? (size_t)0 : (size_t)sn0 * C * 2);      // CT3: BN = Cout, every phase block multiplies by all rows of w2
        const int spy = CT3 ? (sn0 / BN) >> 1 : 0, spx = CT3 ? (sn0 / BN) & 1 : 0;      // (side input: one-phase-per-block form only, see conv_pp_eligible)
#pragma unroll
        for (int i = 0; i < HPW; i++) {
            int piece = wave + 8 * i;
            piece = piece < HALO_PIECES ? piece : HALO_PIECES - 1;
            int hp = piece * 8 + prow_t;
            // chunk swizzle by the halo COLUMN: sw = (hx >> 1) & 7.  A 16-lane ds_read_b128 group covers 16 consecutive columns (split over
            // two image rows for the 16-wide tile); with an even row pitch the LDS half-row bit is hx & 1, so (hx & 1, (hx >> 1) & 7) = hx mod 16
            // is distinct for every lane of the group at every tap shift: conflict-free (the linear-index swizzle was 2-way on 4 lanes)
            // (16x16x32 form: keyed on the column itself, hx & 7 - conflict-free for all three tap shifts of its 16-lane fragment groups, found by search)
            const int sw = ((EPI & 8) ? (hp % HALO_W) : ((hp % HALO_W) >> 1)) & 7;      // padding pixels of the last piece: any consistent value
            hp = hp < HALO_PX ? hp : HALO_PX - 1;
            const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
            int yy = sy0 - 1 + hy, xx = sx0 - 1 + hx;
            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);                    // replicate padding (modules.py:53)
            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            hoff[i] = (unsigned)(((yy * W + xx) * C) * 2 + ((pch_t ^ sw) << 4));
            if constexpr (CT3) hoff2[i] = (unsigned)((((2 * yy + spy) * (2 * W) + 2 * xx + spx) * C2) * 2 + ((pch_t ^ sw) << 4));
        }
    };
    int wrow[NWP];                      // weight-tile row and swizzled chunk offset of this lane in piece i
    unsigned wsw[NWP];
#pragma unroll
    for (int i = 0; i < NWP; i++) {
        wrow[i] = (wave + 8 * i) * 8 + prow;
        wsw[i] = (unsigned)((pch ^ ((wrow[i] >> 1) & 7)) << 4);
    }
    auto issue_halo = [&](int c) {               // chunk c of the conv input, or chunk c - nchunks of the side input
        char* dst = smem + (c & (NH - 1)) * HALO_BYTES;
        const char* src = uniform_ptr(c < nchunks ? in_b + (size_t)c * 128 : in2_b + (size_t)(c - nchunks) * 128);
        const bool side2 = CT3 && c >= nchunks;
#pragma unroll
        for (int i = 0; i < HPW; i++) {
            int piece = wave + 8 * i;
            piece = piece < HALO_PIECES ? piece : HALO_PIECES - 1;
            unsigned off = hoff[i];
            if constexpr (CT3) off = side2 ? hoff2[i] : off;
            __builtin_amdgcn_global_load_lds(CP_GPTR(src + cp_opaque(off)), CP_LPTR(dst + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_w = [&](int kt) {                 // K-step kt = chunk * 9 + tap  ->  weight columns (tap * C + chunk * 64); side steps follow
        char* dst = smem + LDS_W + (kt & 3) * WSLOT;
        const bool side = kt >= nkt1;
        const int c = CT3 ? kt >> 2 : kt / 9, tap = CT3 ? kt & 3 : kt - c * 9;
        const char* src = uniform_ptr(side ? w2_b + (size_t)(kt - nkt1) * 128 : w_b + ((size_t)tap * C + c * 64) * 2);
        const int rowb = side ? C2 * 2 : g.ldw * 2;                  // row pitch of the side weights [N][C2] / conv weights [N][9C] (CT3: [4 Cout][4 Cin])
#pragma unroll
        for (int i = 0; i < NWP; i++) {
            const unsigned off = (unsigned)(wrow[i] * rowb) + wsw[i];
            __builtin_amdgcn_global_load_lds(CP_GPTR(src + cp_opaque(off)), CP_LPTR(dst + (wave + 8 * i) * 1024), 16, 0, 0);
        }
    };

    // ---- fragment addressing ---------------------------------------------------------------------------------------
    // pixel of lane l31 in 32-pixel tile i of this wave: tile-linear index mp = wm*WROWS + i*32 + l31 -> (mp >> 4, mp & 15)
    int hp0[TM], hx0[TM];                 // halo pixel index / halo column of the CENTRE tap
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mp = wm * WROWS + i * 32 + l31;
        hx0[i] = (mp & (TW - 1)) + 1;
        hp0[i] = ((mp >> LOG_TW) + 1) * HALO_W + hx0[i];
    }
    const int sxw = (l31 >> 1) & 7;
    const int w_off = (wn * 64 + l31) * 128 + ((hi ^ sxw) << 4);          // weight row (wn*64 + j*32 + l31); k-step ks: ^ (ks * 32)
    // 16x16x32: pixel block i (one tile row): halo pixel (wm*2*TM + i + 1) * HALO_W + (l15 + 1); weight row wn*64 + j*16 + l15; chunk 4*ks + g4
    const int l15 = lane & 15, g4 = lane >> 4;
    const int w_off16 = (wn * 64 + l15) * 128 + ((g4 ^ ((l15 >> 1) & 7)) << 4);

    f32x16 acc[TM][TN];
    f32x4 acc16[2 * TM][4];
    const int ntot = nchunks + nside;                // halo images consumed: conv chunks, then side chunks
    // one K-step: fragments of (halo image, tap), weights of step kt; DMA for step kt+3 (+ the next halo image at the first step of
    // an image); counted wait; barrier; 16 / 8 MFMAs; barrier.  halo_age: steps since the last halo issue (its 6-10 pieces may
    // still be in flight during the issuing step and the one after).
    int kt = 0, halo_age = 2;
    // The halo fragment addresses depend on (lane, tap) only: left alone, the compiler hoists all of them out of the tile loop (tens of VGPRs)
    // and reloads them from scratch in every K-step - behind vmcnt(0).  l15t is made opaque once per tile so they are recomputed per tap
    // (6 VALU) as in the one-tile-per-workgroup kernel.
    int l15t = l15;
    // 16x16x32 form: the fragment address of (pixel block i, tap (dy, dx), K-step ks) = abase[dx + 1][ks] + (i + 1 + dy) * HALO_W * 128: three
    // column shifts x two K-steps per lane, refreshed once per tile; the row part is a compile-time ds_read offset.  (Round 3: the per-tap
    // recomputation was ~20 VALU in every read segment - issued beside the partner wave's priority-1 MFMAs they cost ~10-20 clocks each and
    // made the READ segment the longer one: 128-channel K-steps 1780 clocks for 2 x 32 MFMAs.)
    int abase[3][2];
    int tpy = 0;                                                          // CT3: row phase of the tile being computed (halo row offset of its taps)
    auto set_abase = [&]() {
        // CT3 phases of this workgroup: BN = Cout -> one phase (py, px) = column block;  BN = 2 Cout (Cout = 64 on the 128-wide form) -> the two phases
        // (py, 0), (py, 1) side by side: a wave's column half wn IS its px (same halo image, fragments shifted by one column), column block = py
        const bool ph2 = CT3 && BN == 2 * g.Cout;
        const int tpx = CT3 ? (ph2 ? wn : (n0 / BN) & 1) : 0;
        if constexpr (CT3) tpy = ph2 ? (n0 / BN) & 1 : (n0 / BN) >> 1;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            // halo column of tap dx = d - 1;  CT3: d = 0, 1 are the phase's two taps (low-res column x + px - 1 + d), d = 2 the centre (side input)
            const int hx = l15t + (CT3 ? (d == 2 ? 1 : tpx + d) : d);
            const int a = wm * (2 * TM * HALO_W * 128) + hx * 128 + ((g4 ^ (hx & 7)) << 4);
            abase[d][0] = a; abase[d][1] = a ^ 64;
        }
    };
    u32x4 sidef[SIDE_REG ? 2 * TM : 1][2];      // SIDE_REG: B fragments of the side map (pixel block i, K-step ks)
    auto kstep = [&](const char* halo, int dy, int dx, bool first_of_image, int image, bool relu, bool from_regs = false) {
        const char* wsl = smem + LDS_W + (kt & 3) * WSLOT;
        u32x4 af[TM][4], wf[TN][4];          // M16: viewed as af16[2*TM][2] / wf16[4][2] (same register count)
        if constexpr (M16) {
            if (from_regs) {
                if constexpr (SIDE_REG) {
#pragma unroll
                    for (int i = 0; i < 2 * TM; i++)
#pragma unroll
                        for (int ks = 0; ks < 2; ks++) af[i >> 1][(i & 1) * 2 + ks] = sidef[i][ks];
                }
            } else {
#pragma unroll
            for (int i = 0; i < 2 * TM; i++) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
                    af[i >> 1][(i & 1) * 2 + ks] = *reinterpret_cast<const u32x4*>(halo + abase[dx + 1][ks] + (i + 1 + dy) * (HALO_W * 128));
            }
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) wf[j >> 1][(j & 1) * 2 + ks] = *reinterpret_cast<const u32x4*>(wsl + (w_off16 ^ (ks * 64)) + j * 2048);
        } else {
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int hp = hp0[i] + dy * HALO_W + dx;
            const int a0 = hp * 128 + ((hi ^ (((hx0[i] + dx) >> 1) & 7)) << 4);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) af[i][ks] = *reinterpret_cast<const u32x4*>(halo + (a0 ^ (ks * 32)));
        }
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) wf[j][ks] = *reinterpret_cast<const u32x4*>(wsl + (w_off ^ (ks * 32)) + j * 4096);
        }
        // DMA: the next halo image (the other halo buffer was last read one barrier ago), weights 3 steps ahead
        const bool halo_now = NH > 1 && first_of_image && image + 1 < ntot;
        if (halo_now) { issue_halo(image + 1); halo_age = 0; }
        const int ahead = nkt - 2 - kt;                    // how many of W(kt+2), W(kt+3) exist
        if (ahead >= 2) issue_w(kt + 3);
        if (halo_now && image >= nchunks) {
            // a side image is consumed in ONE step: the halo just issued is read by the very next step, so it must land now
            // (only the weight pieces issued after it may stay in flight); costs one DMA latency per additional side chunk
            if (ahead >= 2) wait_vm_lgkm<NWP>();
            else wait_vm_lgkm<0>();
        } else if (NH > 1 && halo_age <= 1) {
            if (ahead >= 2) wait_vm_lgkm<2 * NWP + HPW>();
            else if (ahead == 1) wait_vm_lgkm<NWP + HPW>();
            else wait_vm_lgkm<HPW>();
        } else {
            if (ahead >= 2) wait_vm_lgkm<2 * NWP>();
            else if (ahead == 1) wait_vm_lgkm<NWP>();
            else wait_vm_lgkm<0>();
        }
        halo_age++;
        if constexpr (RELU_IN) {
            if (relu) {             // compile-time for the conv taps (always), false for the side input
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) af[i][ks] = relu8(af[i][ks]);
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (M16) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int i = 0; i < 2 * TM; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) mma16<f16>(acc16[i][j], wf[j >> 1][(j & 1) * 2 + ks], af[i >> 1][(i & 1) * 2 + ks]);
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) mma_step<f16>(acc[i][j], wf[j][ks], af[i][ks]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        if (!(grp == 1 && kt == nkt - 1)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        kt++;
    };
    setup(li);
    if constexpr (SIDE_REG) {
        // ordinary loads, issued BEFORE the first DMA: every counted wait below is for something younger, so (vmcnt is in order) they have
        // landed with the tile-head wait; the compiler's own wait in front of their use sits in the last K-step, when nothing else is in flight
#pragma unroll
        for (int i = 0; i < 2 * TM; i++) {
            int y = sy0 + wm * 2 * TM + i, x = sx0 + l15;
            y = y < H ? y : H - 1; x = x < W ? x : W - 1;
            const char* sp = in2_b + ((size_t)y * W + x) * C * 2 + g4 * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) sidef[i][ks] = *reinterpret_cast<const u32x4*>(sp + ks * 64);
        }
    }
    issue_halo(0);
    bool first = true;
    for (;;) {                                      // ======== one tile per iteration ========
    b = sb; y0 = sy0; x0 = sx0; n0 = sn0;
    if constexpr (PERSIST) { asm volatile("" : "+v"(l15t)); if constexpr (M16) set_abase(); }
    else if constexpr (M16) set_abase();
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * TM; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- tile head: the first halo image is already on its way (requested by the previous tile's epilogue / above) -------------------
    if (!first) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // every wave has read its epilogue staging region back: the weight ring is free
        asm volatile("" ::: "memory");
    }
    first = false;
    kt = 0; halo_age = 2;
    issue_w(0);
    issue_w(1);                                   // nkt >= 9
    issue_w(2);
    wait_vm_lgkm<2 * NWP>();                      // in order: the halo image, the previous epilogue's stores and W(0) are behind this
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    for (int c = 0; c < nchunks; c++) {
        const char* halo = smem + (c & (NH - 1)) * HALO_BYTES;
        if constexpr (CT3) {
            const char* hp = halo + tpy * (HALO_W * 128);             // taps (tdy, tdx): halo rows i + py + tdy, columns l15 + px + tdx (abase[tdx])
#pragma unroll
            for (int tap = 0; tap < 4; tap++) kstep(hp, (tap >> 1) - 1, (tap & 1) - 1, tap == 0, c, true);
        } else {
#pragma unroll
        for (int tap = 0; tap < 9; tap++) kstep(halo, tap / 3 - 1, tap % 3 - 1, tap == 0, c, true);
        }
    }
    if constexpr (SIDE_REG) kstep(smem, 0, 0, false, nchunks, false, true);
    else
    for (int c2 = 0; c2 < nside; c2++) kstep(smem + ((nchunks + c2) & (NH - 1)) * HALO_BYTES, 0, CT3 ? 1 : 0, true, nchunks + c2, false);

    // ---- epilogue: bias / uv / ReLU in registers, transpose through LDS, (residual add,) 16-byte pixel-row stores -------------
    // Staging region of this wave: in the WEIGHT RING (every wave has passed the last barrier: all fragment reads are done), so that both halo
    // buffers are free for the next tile's first image while this epilogue runs.
    char* R = smem + (PERSIST ? LDS_W : 0) + wave * (WROWS * 128);
    int rr = lane >> 3, cc = lane & 7;
    if constexpr (PERSIST) asm volatile("" : "+v"(rr), "+v"(cc));          // (per tile: keeps the store-loop address arithmetic out of loop-invariant registers)
    const int nw = n0 + wn * 64;
    const float lo = g.act == ACT_RELU ? 0.f : -3.0e38f;        // branch-free optional ReLU
    const bool has_bias = g.bias != nullptr;
    f16* const outp = reinterpret_cast<f16*>(g.out);
    const f16* const addp = reinterpret_cast<const f16*>(g.add);
    // (1) The compiler waits vmcnt(0) for an ordinary load while LDS-DMA is in flight, so nothing may be loaded behind the halo prefetch
    //     further down: the per-channel vectors are loaded (and used) by the arithmetic in front of it, the residual rows (x + conv(x),
    //     modules.py:66) are requested here - all of them before the first store as well: vmcnt counts stores too and retires in order.
    f16x8 addv[WROWS / 8];
    auto load_skip_rows = [&]() {
        if constexpr (!CONVT) {
            if (addp) {
    #pragma unroll
                for (int it = 0; it < WROWS / 8; it++) {
                    const int mp = wm * WROWS + it * 8 + rr;
                    int y = y0 + (mp >> LOG_TW), x = x0 + (mp & (TW - 1));
                    y = y < H ? y : H - 1; x = x < W ? x : W - 1;
                    addv[it] = *reinterpret_cast<const f16x8*>(addp + (((size_t)b * H + y) * W + x) * g.ldadd + nw + cc * 8);
                }
            }
        }
    };
    if constexpr (PERSIST) load_skip_rows();         // early: their latency passes under the arithmetic below (they must be in before the prefetch)
    u32x4 dota[DOT ? 3 : 1], dotb[DOT ? 2 * TM : 1][2];
    if constexpr (DOT) {
#pragma unroll
        for (int gq = 0; gq < 3; gq++)
            dota[gq] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(g.dot_tab) + (gq < g.dot_nd ? gq : 0) * 1024 + lane * 16);
    }
    if constexpr (M16) {
        // 16x16x32 accumulators: block i = tile row wm*2*TM + i, lane: pixel x0 + l15, channels nw + jj*16 + 4*g4 .. +3
        float u0[2 * TM], u1[2 * TM], v0[2 * TM], v1[2 * TM];
        if constexpr (HAS_UV) {
#pragma unroll
            for (int i = 0; i < 2 * TM; i++) {
                int y = y0 + wm * 2 * TM + i, x = x0 + l15;
                y = y < H ? y : H - 1; x = x < W ? x : W - 1;
                if constexpr (CONVT) {
                    u0[i] = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, 2 * W, 2 * x);
                    u1[i] = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, 2 * W, 2 * x + 1);
                    v0[i] = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, 2 * H, 2 * y);
                    v1[i] = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, 2 * H, 2 * y + 1);
                } else {
                    u0[i] = u1[i] = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, W, x);
                    v0[i] = v1[i] = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, H, y);
                }
            }
        }
        // all four bias vectors requested together and unconditionally (conv_pp_eligible requires a bias): a load under `if (has_bias)` is a
        // branch + vmcnt(0) per vector - four dependent L2 round trips in every tile's epilogue (cdna guide 5, ".s-level traps" (c))
        f32x4 bvs[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) bvs[jj] = *reinterpret_cast<const f32x4*>(g.bias + nw + jj * 16 + 4 * g4);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int n = nw + jj * 16 + 4 * g4;
            const f32x4 bv = bvs[jj];
            f32x4 wu = {0.f, 0.f, 0.f, 0.f}, wv = wu;
            int pdy = 0, pdx = 0;
            if constexpr (HAS_UV) {
                int nco = n;
                if constexpr (CONVT) { const int qd = n / g.Cout; nco = n - qd * g.Cout; pdy = qd >> 1; pdx = qd & 1; }
                wu = *reinterpret_cast<const f32x4*>(g.uv.wu + nco);
                wv = *reinterpret_cast<const f32x4*>(g.uv.wv + nco);
            }
#pragma unroll
            for (int i = 0; i < 2 * TM; i++) {
                const int row = i * 16 + l15;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc16[i][jj][e] + bv[e];
                if constexpr (HAS_UV) {
                    const float uu = pdx ? u1[i] : u0[i];
                    const float vq = pdy ? v1[i] : v0[i];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] += wu[e] * uu + wv[e] * vq;
                }
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], lo);
                const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                if constexpr (DOT) {
                    // B operand of the output-conv MFMA of (block i, phase jj >> 1): halves 0-3 = channels 4*g4 + e (jj even), 4-7 = 16 + 4*g4 + e (jj odd)
                    const u32x2 hw = __builtin_bit_cast(u32x2, hv);
                    dotb[i][jj >> 1][(jj & 1) * 2] = hw[0];
                    dotb[i][jj >> 1][(jj & 1) * 2 + 1] = hw[1];
                } else {
                    *reinterpret_cast<f16x4*>(R + row * 128 + ((((jj * 2 + (g4 >> 1)) ^ (row & 7)) << 4) | ((g4 & 1) << 3))) = hv;
                }
            }
        }
        if constexpr (DOT) {
            // one small MFMA per (group, pixel block, phase); lane group g4 stores pixel block i = g4 (every group holds all four results):
            // pixel (y0 + wm*2*TM + i, x0 + l15), phases (dy = wn, dx = p) = two adjacent high-res pixels, contiguous over the 16 lanes of a row.
            // Stored at once (one result live at a time: the 3 x 4 x 2 results together do not fit beside the kernel's other registers)
            float* const dout = g.dot_out;
            const int nd4 = 4 * g.dot_nd;
#pragma unroll
            for (int i = 0; i < 2 * TM; i++) {
                const int y = y0 + wm * 2 * TM + i, x = x0 + l15;
                const bool mine = g4 == i && y < H && x < W;
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const size_t px = ((size_t)b * 2 * H + 2 * y + wn) * (2 * W) + 2 * x + p;
#pragma unroll
                    for (int gq = 0; gq < 3; gq++)
                        if (gq < g.dot_nd) {
                            f32x4 d = {0.f, 0.f, 0.f, 0.f};
                            mma16<f16>(d, dota[gq], dotb[i][p]);
                            if (mine) *reinterpret_cast<f32x4*>(dout + px * nd4 + 4 * gq) = d;
                        }
                }
            }
        }
    } else {
    float u0[TM], u1[TM], v0[TM], v1[TM];          // separate arrays: a [TM][2] array indexed by the parity goes to scratch
    if constexpr (HAS_UV) {
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int mp = wm * WROWS + i * 32 + l31;
            int y = y0 + (mp >> LOG_TW), x = x0 + (mp & (TW - 1));
            y = y < H ? y : H - 1; x = x < W ? x : W - 1;
            if constexpr (CONVT) {
                // high-res coordinates of the two output parities
                u0[i] = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, 2 * W, 2 * x);
                u1[i] = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, 2 * W, 2 * x + 1);
                v0[i] = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, 2 * H, 2 * y);
                v1[i] = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, 2 * H, 2 * y + 1);
            } else {
                u0[i] = u1[i] = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, W, x);
                v0[i] = v1[i] = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, H, y);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n = nw + j * 32 + 8 * q + 4 * hi;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (has_bias) bv = *reinterpret_cast<const f32x4*>(g.bias + n);
            f32x4 wu = {0.f, 0.f, 0.f, 0.f}, wv = wu;
            int pdy = 0, pdx = 0;
            if constexpr (HAS_UV) {
                int nco = n;
                if constexpr (CONVT) { const int qd = n / g.Cout; nco = n - qd * g.Cout; pdy = qd >> 1; pdx = qd & 1; }
                wu = *reinterpret_cast<const f32x4*>(g.uv.wu + nco);
                wv = *reinterpret_cast<const f32x4*>(g.uv.wv + nco);
            }
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int row = i * 32 + l31;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * q + e] + bv[e];
                if constexpr (HAS_UV) {
                    const float uu = pdx ? u1[i] : u0[i];
                    const float vq = pdy ? v1[i] : v0[i];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] += wu[e] * uu + wv[e] * vq;
                }
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], lo);
                const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4*>(R + row * 128 + ((((j * 4 + q) ^ (row & 7)) << 4) | (hi << 3))) = hv;
            }
        }
    }
    // (2) the accumulators are in the staging region (registers free): the next tile's coordinates and halo offsets, then its first halo image
    if constexpr (!PERSIST) load_skip_rows();        // one tile per workgroup: where the loads always were (fewer live registers in the arithmetic)
    bool more = false;
    if constexpr (PERSIST) {
        li += wgs_x;
        more = li < cnt;
        if (more) setup(li);
        __builtin_amdgcn_s_waitcnt(0x0F70);          // a real S_WAITCNT vmcnt(0) the compiler accounts for: the loads above are complete
        if (more) issue_halo(0);
        asm volatile("" ::: "memory");
    }
    if constexpr (!DOT)
#pragma unroll
    for (int it = 0; it < WROWS / 8; it++) {
        const int row = it * 8 + rr;
        const int mp = wm * WROWS + row;
        const int y = y0 + (mp >> LOG_TW), x = x0 + (mp & (TW - 1));
        u32x4 v = *reinterpret_cast<const u32x4*>(R + row * 128 + ((cc ^ (row & 7)) << 4));
        if (y < H && x < W) {
            if constexpr (CONVT) {
                // 64 staged channels = quadrants (nw/Cout + cc>>2 ...) of Cout = 32 channels, or part of one quadrant when Cout >= 64
                const int n = nw + cc * 8;
                const int qd = n / g.Cout, co = n - qd * g.Cout, pdy = qd >> 1, pdx = qd & 1;
                const size_t idx = ((((size_t)b * 2 * H + 2 * y + pdy) * (2 * W)) + 2 * x + pdx) * g.Cout + co;
                *reinterpret_cast<u32x4*>(outp + idx) = v;
            } else {
                const size_t idx = (((size_t)b * H + y) * W + x) * g.ldc + nw + cc * 8;
                if (addp) {
                    f16x8 h = __builtin_bit_cast(f16x8, v);
                    h += addv[it];                           // fp16 + fp16, as the reference's .half() path does (x + conv(x))
                    v = __builtin_bit_cast(u32x4, h);
                }
                *reinterpret_cast<u32x4*>(outp + idx) = v;
            }
        }
    }
    if (!more) break;
    }                                               // ======== next tile ========
}

template <int BN, int TW, int NH, int EPI>
int launch_conv_cfg(const GemmArgs& g, hipStream_t st) {
    constexpr int smem = NH * Halo<TW>::BYTES + 4 * BN * 128;
    constexpr auto kern = conv_pp_kernel<BN, TW, NH, EPI>;
    if (int rc = set_dyn_lds<kern>(smem)) return rc;
    const long B = (long)g.M / ((long)g.H * g.W);
    const long tiles = B * ((g.H + 15) / 16) * ((g.W + TW - 1) / TW) * (g.N / BN);
    // persistent: as many workgroups as the chip holds at once (LDS: two per CU for the single-image 64-channel form, one otherwise)
    const int ncu = pp_device_cus();
    long slots = (BN == 64 && NH == 1) ? tiles : (long)ncu;          // (the two-per-CU form: one tile per workgroup, see PERSIST)
    const long cap = moge_tune_get("CONV_GRID", 0);      // tests: a small grid makes small problems walk many tiles per workgroup
    if (cap > 0 && !(BN == 64 && NH == 1)) slots = cap;
    const long grid = tiles < slots ? tiles : slots;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, g);
    return (int)hipGetLastError();
}

}  // namespace

// AMODE_CONV3 problems the halo kernel takes (f16): Cin multiple of 64, N = 64 or a multiple of 128, plain / pixel-shuffle store
bool conv_pp_eligible(const GemmArgs& g) {
    if (g.ct3) {            // fused ConvTranspose2d + 3x3 (see the kernel): one phase per column block, BN = Cout
        if ((g.C & 63) || g.K != 4 * g.C || g.ldw != 4 * g.C || !g.bias || g.epi != EPI_CONVT || g.N != 4 * g.Cout) return false;
        if (g.Cout != 128 && g.Cout != 64) return false;
        if (g.add || g.act != ACT_NONE || g.relu_in || g.dot_tab || (g.a2 && !g.w2)) return false;
        if (g.a2 && g.Cout == 64) return false;      // Cout = 64 runs two phases per 128-wide column block: the side image (one high-res pixel per low-res pixel) would differ per wave column
        if (g.H < 1 || g.W < 1 || (long)g.M % ((long)g.H * g.W) != 0) return false;
        if ((long)g.H * g.W * g.C * 2 >= (1L << 31) || (long)g.H * g.W * g.Cout * 8 >= (1L << 31)) return false;      // 32-bit halo offsets (input / side map)
        return true;
    }
    if ((g.C & 63) || g.K != 9 * g.C || g.ldw != 9 * g.C || !g.bias) return false;
    if (g.N != 64 && (g.N & 127)) return false;
    if (g.H < 1 || g.W < 1 || (long)g.M % ((long)g.H * g.W) != 0) return false;
    if ((long)g.H * g.W * g.C * 2 >= (1L << 31)) return false;                     // 32-bit halo offsets
    if (g.a2 && (!g.w2 || g.epi != EPI_STORE)) return false;
    if (g.relu_in && (g.uv.wu || g.epi != EPI_STORE)) return false;
    if (g.epi == EPI_STORE)        // (the residual add is applied after the activation here: never combined by the decoder)
        return (g.ldc & 7) == 0 && (!g.add || ((g.ldadd & 7) == 0 && g.act == ACT_NONE)) && (!g.uv.wu || g.bias) && g.act != ACT_GELU;
    if (g.dot_tab && !(g.epi == EPI_CONVT && g.Cout == 32 && g.N == 128 && g.C == 64 && g.dot_out && g.dot_nd >= 1 && g.dot_nd <= 3 && !g.a2)) return false;
    if (g.epi == EPI_CONVT) return !g.add && g.act == ACT_NONE && (g.Cout == 32 || (g.Cout & 63) == 0) && g.N == 4 * g.Cout;
    return false;
}

template <int BN, int TW, int NH>
static int launch_conv_epi(const GemmArgs& g, hipStream_t st) {
    const int e = (g.uv.wu ? 1 : 0) | (g.epi == EPI_CONVT ? 2 : 0) | (g.relu_in ? 4 : 0);
    if constexpr (BN == 64 && TW == 16 && NH == 1) {
        if (g.a2) return e == 0 ? launch_conv_cfg<BN, TW, NH, 8 | 16>(g, st) : -1;      // register side input (plain store only: what the heads use)
    }
    if constexpr (BN == 128 && TW == 16 && NH == 1) {
        if (g.dot_tab) {                                                                // fused output conv of level 4
            if (e == 2) return launch_conv_cfg<BN, TW, NH, 8 | 2 | 32>(g, st);
            if (e == 3) return launch_conv_cfg<BN, TW, NH, 8 | 2 | 1 | 32>(g, st);
            return -1;
        }
    }
#ifdef MOGE_EXPERIMENTS
    if (TW == 32 || moge_tune_get("CONV_M16", 1) == 0) {       // the v_mfma_f32_32x32x16_f16 form (tools/kbench A-B only)
        switch (e) {
        case 0: return launch_conv_cfg<BN, TW, NH, 0>(g, st);
        case 1: return launch_conv_cfg<BN, TW, NH, 1>(g, st);
        case 2: return launch_conv_cfg<BN, TW, NH, 2>(g, st);
        case 3: return launch_conv_cfg<BN, TW, NH, 3>(g, st);
        case 4: return launch_conv_cfg<BN, TW, NH, 4>(g, st);
        default: return -1;
        }
    }
#endif
    if constexpr (TW == 16) {
        switch (e) {                                             // EPI bit 3: v_mfma_f32_16x16x32_f16
        case 0: return launch_conv_cfg<BN, TW, NH, 8>(g, st);
        case 1: return launch_conv_cfg<BN, TW, NH, 9>(g, st);
        case 2: return launch_conv_cfg<BN, TW, NH, 10>(g, st);
        case 3: return launch_conv_cfg<BN, TW, NH, 11>(g, st);
        case 4: return launch_conv_cfg<BN, TW, NH, 12>(g, st);     // ReLU prologue: plain store only (residual blocks, modules.py:52,58)
        default: return -1;
        }
    }
    return -1;
}


// ------------------------------------------------------------------------------------------------------------------------
// CT3 border (round 6).  The composed conv of conv_pp_kernel<..., CT3> indexes the LOW-res map with clamped coordinates, which on the high-res side is a
// REFLECTING pad (row -1 -> row 1: low-res row 0 with the transposed conv's sy = 1), where the reference pads the ConvTranspose2d OUTPUT by replication
// (row -1 -> row 0, modules.py:163).  The two agree everywhere except on the outermost ring of output pixels, where
//     exact - composed = sum over the 3x3 taps (ky, kx) that leave the image of  P(ky, kx; parity of the replicated coordinate) . L(its cell)
//                                                                               - P(ky, kx; parity of the reflected coordinate) . L(its cell),
// P(ky, kx; sy, sx) = W3[:, :, ky, kx] . WT[:, :, sy, sx]^T.  Per border class (4 edges x 2 parities along the edge + 4 corners) these sums involve at most two
// low-res cells and are pre-summed at pack time into dw[class][co][slot * Cin + ci]; this kernel evaluates them as small MFMA GEMMs (16 border pixels per
// wave, A fragments straight from global: the weights are shared by every wave, 4 W + 4 H - 4 pixels per image) and adds the result to the stored output.
// Classes: 0 / 1 top, px = 0 / 1 (cells (0, x + px - 1), (0, x + px));  2 / 3 bottom;  4 / 5 left, py = 0 / 1 (cells (y + py - 1, 0), (y + py, 0));  6 / 7 right;
// 8 ... 11 corners TL TR BL BR (one cell).
// ------------------------------------------------------------------------------------------------------------------------
template <int NJ>      // Cout / 16
__global__ __launch_bounds__(256) void ct3_border_kernel(const f16* __restrict__ in, const f16* __restrict__ dw, f16* __restrict__ out, int B, int H, int W, int Cin) {
    constexpr int Cout = NJ * 16;
    // 16 border pixels per WORKGROUP: its four waves split the K-steps (slot, 32-channel chunk) round-robin and wave 0 adds the partial sums in wave order (fixed:
    // deterministic, the same for every batch size).  One image has 4 W + 4 H - 4 = 476 such pixels at the 120^2 level: with 16 pixels per WAVE over the whole K
    // (first form) a launch was 24 workgroups walking a chain of 16 dependent load -> MFMA steps, 20-29 us on the one-image critical path.
    __shared__ f32x4 red[3][NJ][64];
    const int cls = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int n_c = cls < 4 ? W - 1 : (cls < 8 ? H - 1 : 1);
    const long Mc = (long)B * n_c;
    const long m0 = (long)blockIdx.x * 16;
    if (m0 >= Mc) return;
    long m = m0 + l15;
    const bool valid = m < Mc;
    m = valid ? m : Mc - 1;
    const int b = (int)(m / n_c), idx = (int)(m - (long)b * n_c);
    int Y, X, cy0, cx0, cy1, cx1;
    if (cls < 4) {                                   // top / bottom edge
        const int px = cls & 1, x = px ? idx : idx + 1;
        Y = cls < 2 ? 0 : 2 * H - 1; X = 2 * x + px;
        cy0 = cy1 = cls < 2 ? 0 : H - 1; cx0 = x + px - 1; cx1 = x + px;
    } else if (cls < 8) {                            // left / right edge
        const int py = cls & 1, y = py ? idx : idx + 1;
        X = cls < 6 ? 0 : 2 * W - 1; Y = 2 * y + py;
        cx0 = cx1 = cls < 6 ? 0 : W - 1; cy0 = y + py - 1; cy1 = y + py;
    } else {                                         // corners
        Y = (cls & 2) ? 2 * H - 1 : 0; X = (cls & 1) ? 2 * W - 1 : 0;
        cy0 = cy1 = (cls & 2) ? H - 1 : 0; cx0 = cx1 = (cls & 1) ? W - 1 : 0;
    }
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f16* wbase = dw + ((size_t)cls * Cout + l15) * (2 * Cin) + 8 * g4;
    const int nslot = cls < 8 ? 2 : 1;
    const int nkc = Cin >> 5, nsteps = nslot * nkc;
    for (int stp = wave; stp < nsteps; stp += 4) {
        const int slot = stp >= nkc ? 1 : 0, kc = (stp - slot * nkc) << 5;
        const f16* pp = in + (((size_t)b * H + (slot ? cy1 : cy0)) * W + (slot ? cx1 : cx0)) * Cin + 8 * g4;
        const f16* wp = wbase + slot * Cin;
        const u32x4 pf = *reinterpret_cast<const u32x4*>(pp + kc);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const u32x4 wf = *reinterpret_cast<const u32x4*>(wp + (size_t)j * 16 * (2 * Cin) + kc);
            mma16<f16>(acc[j], wf, pf);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < NJ; j++) red[wave - 1][j][lane] = acc[j];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; w++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const f32x4 p = red[w][j][lane];
#pragma unroll
            for (int e = 0; e < 4; e++) acc[j][e] += p[e];
        }
    if (!valid) return;
    f16* op = out + (((size_t)b * 2 * H + Y) * (2 * W) + X) * Cout + 4 * g4;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        f16x4 v = *reinterpret_cast<const f16x4*>(op + j * 16);
        v = f16x4{(f16)((float)v[0] + acc[j][0]), (f16)((float)v[1] + acc[j][1]), (f16)((float)v[2] + acc[j][2]), (f16)((float)v[3] + acc[j][3])};
        *reinterpret_cast<f16x4*>(op + j * 16) = v;
    }
}

int launch_ct3_border(const void* in, const void* dw, void* out, int B, int H, int W, int Cin, int Cout, hipStream_t st) {
    if ((Cin & 31) || (Cout != 128 && Cout != 64) || H < 1 || W < 1) return -1;
    const long mmax = (long)B * ((W > H ? W : H) - 1 > 1 ? (W > H ? W : H) - 1 : 1);
    const dim3 grid((unsigned)((mmax + 15) / 16), 12);
    if (Cout == 128) hipLaunchKernelGGL(ct3_border_kernel<8>, grid, dim3(256), 0, st, (const f16*)in, (const f16*)dw, (f16*)out, B, H, W, Cin);
    else hipLaunchKernelGGL(ct3_border_kernel<4>, grid, dim3(256), 0, st, (const f16*)in, (const f16*)dw, (f16*)out, B, H, W, Cin);
    return (int)hipGetLastError();
}

int launch_conv_pp(const GemmArgs& g, hipStream_t st) {
    if (g.ct3) {
        // one kernel form (128 output columns per workgroup: 64 px x 64 ch per wave): Cout = 128 -> one phase per column block, Cout = 64 -> two
        return g.uv.wu ? launch_conv_cfg<128, 16, 2, 8 | 2 | 64 | 1>(g, st) : launch_conv_cfg<128, 16, 2, 8 | 2 | 64>(g, st);
    }
    // a single halo image: one buffer.  A 64-channel layer with a side input runs on the single-buffer form too (side fragments in registers)
    const bool side_reg = g.C == 64 && g.N == 64 && g.a2 && !g.uv.wu && !g.relu_in && g.epi == EPI_STORE && moge_tune_get("CONV_SIDE_REG", 1);
    const bool one_image = g.C == 64 && (!g.a2 || side_reg);
    if (g.N == 64) {
        // 32-pixel-wide tiles (64 px x 64 ch per wave) when the image is wide enough to fill them
#ifdef MOGE_EXPERIMENTS
        if (one_image && g.W >= 32 && moge_tune_get("CONV_TW32", 0)) return launch_conv_epi<64, 32, 1>(g, st);
#endif
        return one_image ? launch_conv_epi<64, 16, 1>(g, st) : launch_conv_epi<64, 16, 2>(g, st);
    }
    return one_image ? launch_conv_epi<128, 16, 1>(g, st) : launch_conv_epi<128, 16, 2>(g, st);
}
