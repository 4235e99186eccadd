// Fused residual block of the decoder (modules.py:47-68 with norms = Identity), fp16 throughput path for gfx950, C = 64:
//
//   out = x + conv2(relu(conv1(relu(x)) + b1)) + b2            (both convs 3x3, replicate padding, NHWC fp16, fp32 accumulate)
//
// conv_pp.hip runs this as two launches: x -> h, h -> y, + x = five passes over a map of which two are compulsory (the 64-channel level
// at 480 x 480 is 29.5 MB per image: its convs are HBM-side bound, rocprof r02zl: 0.94 GB in + 0.94 GB out per launch, MFMA busy 0.35).
// Here ONE persistent workgroup per CU walks 16 x 16 output tiles and keeps the intermediate tile in LDS:
//
//   in-halo  20 x 20 pixels x 64 ch  (LDS-DMA, clamped source indices = replicate padding of x)                         50 KiB (+2 pad)
//   mid      18 x 18 pixels x 64 ch  = relu(conv1 + b1) in fp16 (exactly what the two-launch path stores), row pitch 20      46 KiB
//   weights  [64][64] per tap, 4-slot ring, three K-steps ahead, conv1's nine taps then conv2's nine                          32 KiB
//
// conv1 is evaluated on the 18 x 18 halo of the output tile (1.27 x its FLOPs; 23 MFMA pixel blocks of 16 over the pitch-20 index, three per
// wave) so that conv2 needs nothing from neighbouring tiles.  REPLICATE PADDING OF h: a mid pixel outside the image is never used - conv2
// clamps the image coordinate of every tap before it addresses the mid tile (an out-of-image mid position would hold conv1 evaluated THERE,
// which is not h at the clamped position).  Schedule per tile: [conv1: 9 K-steps on the in-halo] -> mid written from registers -> the next
// tile's in-halo requested (the buffer is dead: it lands under conv2) -> [conv2: 9 K-steps on mid] -> epilogue staged through the dead mid
// buffer, residual rows re-read from L2, next tile's first three weight steps requested in front of the stores.  K-steps as in conv_pp.hip:
// two wave groups half a step apart, counted vmcnt (in order over DMA, loads and stores on gfx950), raw barriers.
// Chunk swizzle of both pixel images: chunk ^ (q & 7), q = linear pixel index at pitch 20: a 16-lane ds_read_b128 group reads 16 consecutive
// q at every tap shift = 16 distinct (half-row, chunk) slots (exhaustive check over all bases: tools/lds_swizzle_check.py).
#include "common.h"

#define RB_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define RB_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

constexpr int RB_PITCH = 20;                      // pixels per row of both LDS images
constexpr int RB_IN_PIECES = 50;                  // 400 pixels / 8 per 1-KiB DMA piece
constexpr int RB_IN_BYTES = 52 * 1024;            // + 2 pieces only the padding lanes of the last mid block read
constexpr int RB_MID_BLOCKS = 23;                 // 18 rows x pitch 20 = 360 positions -> 23 blocks of 16 (368)
constexpr int RB_MID_BYTES = RB_MID_BLOCKS * 16 * 128;
constexpr int RB_WSLOT = 64 * 128;
constexpr int RB_LDS_MID = RB_IN_BYTES;
constexpr int RB_LDS_W = RB_IN_BYTES + RB_MID_BYTES;
constexpr int RB_SMEM = RB_LDS_W + 4 * RB_WSLOT;  // 133120 B: one workgroup per CU
constexpr int RB_HPW = (RB_IN_PIECES + 7) / 8;    // halo pieces per wave (surplus pieces repeat the last one)
constexpr int RB_NST = 4;                         // epilogue stores per lane (32 pixel rows / 8 per pass)

template <int N> __device__ __forceinline__ void rb_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ u32x4 rb_relu8(u32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned t = v[i];
        asm("v_pk_max_f16 %0, %0, 0" : "+v"(t));
        v[i] = t;
    }
    return v;
}

// VAR (tools/kbench A-B): bit 0 = no s_setprio around the MFMA segments, bit 1 = relu(x) applied at the head of the MFMA segment instead of
// in the read segment (VALU in a read segment is starved while the partner wave of the SIMD issues MFMAs at priority 1)
template <int VAR>
__global__ __launch_bounds__(512, 2) void conv_rb_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int prow = lane >> 3, pch = lane & 7;
    const int H = g.H, W = g.W;
    const int tx_n = (W + 15) >> 4, ty_n = (H + 15) >> 4;

    // ---- persistent tile walk: XCD-contiguous ranges of the tile order (x fastest, then y, then image), as conv_pp.hip ----------------
    int li, cnt, start, wgs_x;
    {
        const int nwg = gridDim.x;
        const int ntiles = (g.M / (H * W)) * ty_n * tx_n;
        const int nx = nwg < 8 ? nwg : 8, xcd = blockIdx.x % nx;
        const int q = ntiles / nx, r = ntiles % nx;
        cnt = q + (xcd < r ? 1 : 0);
        start = xcd * q + min(xcd, r);
        wgs_x = (nwg - xcd + nx - 1) / nx;
        li = blockIdx.x / nx;
    }
    if (li >= cnt) return;

    // ---- per-channel vectors: loaded ONCE, before any LDS-DMA is in flight (the compiler waits vmcnt(0) for an ordinary load behind a DMA) ---
    // accumulator layout (v_mfma_f32_16x16x32_f16, A = weights, B = pixels): lane holds pixel l15, channels 16*j + 4*g4 + 0..3
    f32x4 b1v[4], b2v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        b1v[j] = *reinterpret_cast<const f32x4*>(g.bias + 16 * j + 4 * g4);
        b2v[j] = *reinterpret_cast<const f32x4*>(g.rb_bias2 + 16 * j + 4 * g4);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): they are in registers before the first DMA

    // ---- DMA sources --------------------------------------------------------------------------------------------------------------------
    const int wrow = wave * 8 + prow;                // weight row (output channel) of this lane's piece
    const unsigned woff = (unsigned)(wrow * (9 * 64 * 2)) + (unsigned)((pch ^ ((wrow >> 1) & 7)) << 4);
    const char* const w1_b = reinterpret_cast<const char*>(g.w);
    const char* const w2_b = reinterpret_cast<const char*>(g.rb_w2);
    const char* in_b;                                // image base of the tile whose in-halo is being requested
    unsigned hoff[RB_HPW];
    int sb, sy0, sx0;
    auto setup = [&](int idx) {
        int prow_t = prow, pch_t = pch;
        asm volatile("" : "+v"(prow_t), "+v"(pch_t));        // per tile: keeps the per-piece coordinates out of loop-invariant registers
        int t = start + idx;
        const int tx = t % tx_n; t /= tx_n;
        const int ty = t % ty_n;
        sb = t / ty_n; sy0 = ty * 16; sx0 = tx * 16;
        in_b = reinterpret_cast<const char*>(g.a) + (size_t)sb * H * W * 128;
#pragma unroll
        for (int i = 0; i < RB_HPW; i++) {
            int piece = wave + 8 * i;
            piece = piece < RB_IN_PIECES ? piece : RB_IN_PIECES - 1;
            const int q = piece * 8 + prow_t;
            const int hy = q / RB_PITCH, hx = q - hy * RB_PITCH;
            int yy = sy0 - 2 + hy, xx = sx0 - 2 + hx;
            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);                    // replicate padding of x (modules.py:53)
            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            hoff[i] = (unsigned)((yy * W + xx) * 128 + ((pch_t ^ (q & 7)) << 4));
        }
    };
    auto issue_halo = [&]() {
        const char* src = uniform_ptr(in_b);
#pragma unroll
        for (int i = 0; i < RB_HPW; i++) {
            int piece = wave + 8 * i;
            piece = piece < RB_IN_PIECES ? piece : RB_IN_PIECES - 1;
            __builtin_amdgcn_global_load_lds(RB_GPTR(src + hoff[i]), RB_LPTR(smem + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_w = [&](int kt) {                     // K-step kt: conv1 tap kt (kt < 9), conv2 tap kt - 9
        const char* src = uniform_ptr((kt < 9 ? w1_b : w2_b) + (kt < 9 ? kt : kt - 9) * 128);
        __builtin_amdgcn_global_load_lds(RB_GPTR(src + woff), RB_LPTR(smem + RB_LDS_W + (kt & 3) * RB_WSLOT + wave * 1024), 16, 0, 0);
    };

    // ---- fragment addressing --------------------------------------------------------------------------------------------------------------
    const int w_off16 = l15 * 128 + ((g4 ^ ((l15 >> 1) & 7)) << 4);       // weight row 16*j + l15, chunk 4*ks + g4 (^ ks*64, + j*2048)
    int l15t = l15;                                   // made opaque per tile (the (lane, tap) addresses are tile-invariant: hoisted they go to scratch)
    u32x4 wf[4][2];
    auto read_w = [&](int kt) {
        const char* wsl = smem + RB_LDS_W + (kt & 3) * RB_WSLOT;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) wf[j][ks] = *reinterpret_cast<const u32x4*>(wsl + (w_off16 ^ (ks * 64)) + j * 2048);
    };
    // the two barriers of a K-step around its MFMAs; group 1 runs one barrier behind group 0 and skips the last one of a conv
    auto pre_mfma = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(1);
    };
    // The hand-over barrier of an MFMA segment may sit E MFMAs BEFORE its end (VAR bits 2-3: E = 4 / 8 / 12): the partner group starts its
    // MFMAs while this group's last ones are still issuing, so the matrix pipe does not idle for the barrier's release latency at every
    // hand-over (measured ~190 clocks each, two per K-step).  No data hazard is attached to the position: MFMAs touch registers only.
    constexpr int EARLY = ((VAR >> 2) & 3) * 4;
    auto handover = [&](bool last) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        if (!(grp == 1 && last)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto post_mfma = [&](bool last) {
        if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);
        if constexpr (EARLY == 0) handover(last);
        else __builtin_amdgcn_sched_barrier(0);
    };
    // optional s_memtime timeline of workgroup 0 (tools/kbench rb with KB_TS): wave 0 and wave 4, eight stamps per tile for the first six tiles
    unsigned long long* const ts = (g.dbg_ts && blockIdx.x == 0 && (wave & 3) == 0) ? g.dbg_ts + (wave >> 2) * 64 : nullptr;
    int ts_tile = 0;
#define RB_STAMP(i) do { if (ts && ts_tile < 6 && lane == 0) ts[ts_tile * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    f32x4 acc1[3][4], acc2[2][4];
    int b, y0, x0;
    setup(li);
    issue_halo();
    issue_w(0); issue_w(1); issue_w(2);
    bool first = true;
    for (;;) {                                        // ======== one tile per iteration ========
        b = sb; y0 = sy0; x0 = sx0;
        RB_STAMP(0);
        asm volatile("" : "+v"(l15t));
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // tile head: in order behind this wait are the in-halo (requested during the previous tile's conv2 / above) and W(0); in front of it
        // W(1), W(2) and the previous epilogue's stores
        if (first) rb_wait<2>(); else rb_wait<2 + RB_NST>();
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        RB_STAMP(1);

        // ---- conv1 on the in-halo: mid position m = 16*blk + l15 (pitch 20), in-halo pixel of tap (dy, dx): q = m + 21 + 20*dy + dx -----------
#pragma unroll
        for (int kt = 0; kt < 9; kt++) {
            const int dy = kt / 3 - 1, dx = kt % 3 - 1;
            u32x4 af[3][2];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                int blk = wave * 3 + i;
                blk = blk < RB_MID_BLOCKS ? blk : RB_MID_BLOCKS - 1;
                const int q = blk * 16 + l15t + 21 + dy * RB_PITCH + dx;
                const int a0 = q * 128 + ((g4 ^ (q & 7)) << 4);
#pragma unroll
                for (int ks = 0; ks < 2; ks++) af[i][ks] = *reinterpret_cast<const u32x4*>(smem + (a0 ^ (ks * 64)));
            }
            if (kt == 4 && ts && ts_tile == 3 && lane == 0) ts[48] = __builtin_readcyclecounter();
            read_w(kt);
            issue_w(kt + 3);                          // kt + 3 <= 11: conv2's first three steps are requested by conv1's last three
            if (kt == 4 && ts && ts_tile == 3 && lane == 0) ts[49] = __builtin_readcyclecounter();
            // W(kt + 1) has landed (in order: everything older too); in flight stay W(kt + 2), W(kt + 3) and, on the first two steps of a
            // tile that follows another, the previous epilogue's stores
            if (kt < 2 && !first) rb_wait<2 + RB_NST>(); else rb_wait<2>();
            if constexpr (!(VAR & 2)) {
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) af[i][ks] = rb_relu8(af[i][ks]);      // relu(x) (modules.py:52)
            }
            if (kt == 4 && ts && ts_tile == 3 && lane == 0) ts[50] = __builtin_readcyclecounter();
            pre_mfma();
            if (kt == 4 && ts && ts_tile == 3 && lane == 0) ts[51] = __builtin_readcyclecounter();
            if constexpr ((VAR & 2) != 0) {
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) af[i][ks] = rb_relu8(af[i][ks]);
            }
#pragma unroll
            for (int n = 0; n < 24; n++) {
                const int ks = n / 12, i = (n % 12) / 4, j = n % 4;
                if (EARLY > 0 && n == 24 - EARLY) handover(kt == 8);
                mma16<f16>(acc1[i][j], wf[j][ks], af[i][ks]);
            }
            if (kt == 4 && ts && ts_tile == 3 && lane == 0) ts[52] = __builtin_readcyclecounter();
            post_mfma(kt == 8);
            if (kt == 4 && ts && ts_tile == 3 && lane == 0) ts[53] = __builtin_readcyclecounter();
        }
        // ---- transition: every wave has read its last in-halo fragment (group 0's last barrier pairs with the one group 1 passes after its
        // step-8 reads): the in-halo buffer and weight slot 0 are free.  Next tile's coordinates, W(12), then the next in-halo image (after the
        // last tile the workgroup re-requests its own, which nobody reads: the request count stays a compile-time constant) -------------------
        RB_STAMP(2);
        bool more;
        {
            const int nli = li + wgs_x;
            more = nli < cnt;
            if (more) { li = nli; setup(li); }
        }
        issue_w(12);
        issue_halo();
        // mid = relu(conv1 + b1) in fp16: what the two-launch path stores between its convs
#pragma unroll
        for (int i = 0; i < 3; i++) {
            int blk = wave * 3 + i;
            blk = blk < RB_MID_BLOCKS ? blk : RB_MID_BLOCKS - 1;
            const int m = blk * 16 + l15t;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(acc1[i][j][e] + b1v[j][e], 0.f);
                const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4*>(smem + RB_LDS_MID + m * 128 + ((((2 * j + (g4 >> 1)) ^ (m & 7)) << 4) | ((g4 & 1) << 3))) = hv;
            }
        }
        // conv2 tap coordinates inside the mid tile: the IMAGE coordinate of the tap is clamped (replicate padding of h), then made tile-relative
        int mrow[2][3], mcol[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                int yy = y0 + 2 * wave + i + d - 1;
                yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                mrow[i][d] = (yy - y0 + 1) * RB_PITCH;
            }
            int xx = x0 + l15t + d - 1;
            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            mcol[d] = xx - x0 + 1;
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's mid rows are written (W(9) landed with step 8's wait)
        __builtin_amdgcn_s_barrier();                          // the whole mid tile is written
        if (grp == 1) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        RB_STAMP(3);

        // ---- conv2 on the mid tile: output rows 2*wave + i, column l15 ------------------------------------------------------------------------
#pragma unroll
        for (int kt = 9; kt < 18; kt++) {
            const int tap = kt - 9, dyi = tap / 3, dxi = tap % 3;
            u32x4 af[2][2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int q = mrow[i][dyi] + mcol[dxi];
                const int a0 = q * 128 + ((g4 ^ (q & 7)) << 4);
#pragma unroll
                for (int ks = 0; ks < 2; ks++) af[i][ks] = *reinterpret_cast<const u32x4*>(smem + RB_LDS_MID + (a0 ^ (ks * 64)));
            }
            read_w(kt);
            if (kt >= 10 && kt + 3 < 18) issue_w(kt + 3);      // (W(12) went out at the transition, in front of the in-halo request)
            // in flight behind W(kt + 1): step 9: W(11) W(12) H;  10: W(12) H W(13);  11: H W(13) W(14);  12..14: two weight steps (the in-halo
            // image has landed with step 12's wait: three and a half steps after its request);  15: W(17);  16, 17: nothing
            if (kt <= 11) rb_wait<2 + RB_HPW>();
            else if (kt <= 14) rb_wait<2>();
            else if (kt == 15) rb_wait<1>();
            else rb_wait<0>();
            pre_mfma();
#pragma unroll
            for (int n = 0; n < 16; n++) {
                const int ks = n / 8, i = (n % 8) / 4, j = n % 4;
                if (EARLY > 0 && n == 16 - EARLY) handover(kt == 17);
                mma16<f16>(acc2[i][j], wf[j][ks], af[i][ks]);
            }
            post_mfma(kt == 17);
        }

        // ---- epilogue: + b2, fp16, transposed through this wave's 4 KiB of the (dead) mid buffer, + x, 16-byte pixel-row stores ------------
        RB_STAMP(4);
        char* R = smem + RB_LDS_MID + wave * 4096;
        int rr = lane >> 3, cc = lane & 7;
        asm volatile("" : "+v"(rr), "+v"(cc));
        const f16* const addp = reinterpret_cast<const f16*>(g.add);
        f16x8 addv[RB_NST];
        unsigned ooff[RB_NST];
#pragma unroll
        for (int it = 0; it < RB_NST; it++) {
            const int row = it * 8 + rr;                       // staged pixel row: tile row 2*wave + (row >> 4), column row & 15
            const int y = y0 + 2 * wave + (row >> 4), x = x0 + (row & 15);
            const int yc = y < H ? y : H - 1, xc = x < W ? x : W - 1;
            addv[it] = *reinterpret_cast<const f16x8*>(addp + (((size_t)b * H + yc) * W + xc) * g.ldadd + cc * 8);   // the skip rows: L2 (this tile's halo)
            ooff[it] = (y < H && x < W) ? (unsigned)(((y * W + x) * g.ldc + cc * 8) * 2) : 0xFFFFFFFFu;              // outside the descriptor: dropped
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int row = i * 16 + l15;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc2[i][j][e] + b2v[j][e];
                const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4*>(R + row * 128 + ((((2 * j + (g4 >> 1)) ^ (row & 7)) << 4) | ((g4 & 1) << 3))) = hv;
            }
        const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(uniform_ptr(reinterpret_cast<const char*>(g.out) + (size_t)b * H * W * g.ldc * 2)), 0, H * W * g.ldc * 2, 0x00020000);
        __builtin_amdgcn_s_waitcnt(0x0F70);          // a real vmcnt(0) the compiler accounts for: the skip rows are in (the in-halo landed long ago)
        RB_STAMP(5);
        if (more) { issue_w(0); issue_w(1); issue_w(2); }      // next tile's first weight steps, in FRONT of the stores (vmcnt is in order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < RB_NST; it++) {
            const int row = it * 8 + rr;
            u32x4 v = *reinterpret_cast<const u32x4*>(R + row * 128 + ((cc ^ (row & 7)) << 4));
            f16x8 h = __builtin_bit_cast(f16x8, v);
            h += addv[it];                                     // fp16 + fp16 as the two-launch path (and the reference's .half() model): x + conv(x), modules.py:66
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), ob, (int)ooff[it], 0, 0);
        }
        RB_STAMP(6);
        ts_tile++;
        if (!more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; }      // no DMA may land in this CU's LDS after the workgroup has left
        first = false;
    }                                                 // ======== next tile ========
}

#undef RB_STAMP
}  // namespace

// fp16, C = Cin = Cout = 64, ReLU prologue, plain residual (add = the block's own input), no uv / side input
bool conv_rb_eligible(const GemmArgs& g) {
    if (g.C != 64 || g.N != 64 || g.K != 9 * 64 || g.ldw != 9 * 64) return false;
    if (!g.rb_w2 || !g.rb_bias2 || !g.bias || !g.add || !g.relu_in) return false;
    if (g.a2 || g.uv.wu || g.epi != EPI_STORE || g.act != ACT_NONE) return false;
    if (g.ldc != 64 || g.ldadd != 64) return false;
    if (g.H < 1 || g.W < 1 || (long)g.M % ((long)g.H * g.W) != 0) return false;
    if ((long)g.H * g.W * 128 >= (1L << 31)) return false;                      // 32-bit offsets inside an image
    return true;
}

template <int VAR>
static int launch_conv_rb_var(const GemmArgs& g, hipStream_t st) {
    if (int rc = set_dyn_lds<conv_rb_kernel<VAR>>(RB_SMEM)) return rc;
    const long B = (long)g.M / ((long)g.H * g.W);
    const long tiles = B * ((g.H + 15) / 16) * ((g.W + 15) / 16);
    long slots = pp_device_cus();
    const long cap = moge_tune_get("CONV_GRID", 0);      // tests: a small grid makes small problems walk many tiles per workgroup
    if (cap > 0) slots = cap;
    const long grid = tiles < slots ? tiles : slots;
    hipLaunchKernelGGL(conv_rb_kernel<VAR>, dim3((unsigned)grid), dim3(512), RB_SMEM, st, g);
    return (int)hipGetLastError();
}

int launch_conv_rb(const GemmArgs& g, hipStream_t st) {
    if (!conv_rb_eligible(g)) return -1;
    switch (moge_tune_get("CONV_RB_VAR", 0)) {
    case 1: return launch_conv_rb_var<1>(g, st);
    case 2: return launch_conv_rb_var<2>(g, st);
    case 3: return launch_conv_rb_var<3>(g, st);
    case 4: return launch_conv_rb_var<4>(g, st);
    case 8: return launch_conv_rb_var<8>(g, st);
    case 12: return launch_conv_rb_var<12>(g, st);
    default: return launch_conv_rb_var<0>(g, st);
    }
}
