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
// (in place allowed), and the pixel-shuffle store of the 4-phase "bilinear x2 + 3x3" resampler (EPI_CONVT).
#include "common.h"

#define CP_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define CP_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

constexpr int HALO_W = 18;                       // 16 + 2
constexpr int HALO_PX = HALO_W * HALO_W;         // 324
constexpr int HALO_PIECES = 41;                  // ceil(324 / 8) pieces of 8 pixels x 128 B
constexpr int HALO_BYTES = HALO_PIECES * 1024;   // 41984
constexpr int HPW = 6;                           // halo pieces per wave (8 waves x 6 >= 41; surplus pieces repeat piece 40)

__device__ __forceinline__ u32x4 relu8(u32x4 v) {
    f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = h[i] > (f16)0 ? h[i] : (f16)0;
    return __builtin_bit_cast(u32x4, h);
}

template <int N> __device__ __forceinline__ void wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// BN = 128: waves 4 (pixels) x 2 (channels), 64 px x 64 ch per wave;  BN = 64: waves 8 x 1, 32 px x 64 ch per wave
// NH = halo buffers (1 when Cin == 64: a single chunk);  EPI: bit 0 = uv term, bit 1 = pixel-shuffle (EPI_CONVT) store
template <int BN, int NH, int EPI>
__global__ __launch_bounds__(512, (BN == 64 && NH == 1) ? 4 : 2) void conv_pp_kernel(const GemmArgs g) {
    constexpr bool CONVT = (EPI & 2) != 0, HAS_UV = (EPI & 1) != 0;
    constexpr int WN = BN / 64, WM = 8 / WN, TM = 8 / WM, TN = 2;       // TM 32-pixel tiles per wave
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
    const int tx_n = (W + 15) >> 4, ty_n = (H + 15) >> 4;
    const int nbn = g.N / BN;
    int wg;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int bn = wg % nbn;
    int t = wg / nbn;
    const int tx = t % tx_n; t /= tx_n;
    const int ty = t % ty_n;
    const int b = t / ty_n;
    const int y0 = ty * 16, x0 = tx * 16, n0 = bn * BN;
    const int nchunks = NH == 1 ? 1 : (C >> 6);        // NH == 1: Cin == 64, straight-line 9-step K loop
    const int nkt = nchunks * 9;

    // ---- DMA sources ----------------------------------------------------------------------------------------------
    const char* in_b = reinterpret_cast<const char*>(g.a) + (size_t)b * H * W * C * 2;
    const int prow = lane >> 3, pch = lane & 7;
    unsigned hoff[HPW];                 // byte offset of this lane's source chunk for halo piece i (chunk 0 of Cin)
#pragma unroll
    for (int i = 0; i < HPW; i++) {
        int piece = wave + 8 * i;
        piece = piece < HALO_PIECES ? piece : HALO_PIECES - 1;
        int hp = piece * 8 + prow;
        const int sw = (hp >> 1) & 7;                                   // swizzle follows the LDS pixel index, also for the padding pixels
        hp = hp < HALO_PX ? hp : HALO_PX - 1;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);                    // replicate padding (modules.py:53)
        xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
        hoff[i] = (unsigned)(((yy * W + xx) * C) * 2 + ((pch ^ sw) << 4));
    }
    const char* w_b = reinterpret_cast<const char*>(g.w) + (size_t)n0 * g.ldw * 2;
    unsigned woff[NWP];
#pragma unroll
    for (int i = 0; i < NWP; i++) {
        const int row = (wave + 8 * i) * 8 + prow;
        woff[i] = (unsigned)(row * g.ldw * 2 + ((pch ^ ((row >> 1) & 7)) << 4));
    }
    auto issue_halo = [&](int c) {
        char* dst = smem + (c & (NH - 1)) * HALO_BYTES;
        const char* src = uniform_ptr(in_b + (size_t)c * 128);
#pragma unroll
        for (int i = 0; i < HPW; i++) {
            int piece = wave + 8 * i;
            piece = piece < HALO_PIECES ? piece : HALO_PIECES - 1;
            __builtin_amdgcn_global_load_lds(CP_GPTR(src + hoff[i]), CP_LPTR(dst + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_w = [&](int kt) {                 // K-step kt = chunk * 9 + tap  ->  weight columns (tap * C + chunk * 64)
        const int c = kt / 9, tap = kt - c * 9;
        char* dst = smem + LDS_W + (kt & 3) * WSLOT;
        const char* src = uniform_ptr(w_b + ((size_t)tap * C + c * 64) * 2);
#pragma unroll
        for (int i = 0; i < NWP; i++)
            __builtin_amdgcn_global_load_lds(CP_GPTR(src + woff[i]), CP_LPTR(dst + (wave + 8 * i) * 1024), 16, 0, 0);
    };

    // ---- fragment addressing ---------------------------------------------------------------------------------------
    // pixel of lane l31 in 32-pixel tile i of this wave: tile-linear index mp = wm*WROWS + i*32 + l31 -> (mp >> 4, mp & 15)
    int hp0[TM];                          // halo pixel index of the CENTRE tap
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mp = wm * WROWS + i * 32 + l31;
        hp0[i] = ((mp >> 4) + 1) * HALO_W + (mp & 15) + 1;
    }
    const int sxw = (l31 >> 1) & 7;
    const int w_off = (wn * 64 + l31) * 128 + ((hi ^ sxw) << 4);          // weight row (wn*64 + j*32 + l31); k-step ks: ^ (ks * 32)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // ---- prologue -----------------------------------------------------------------------------------------------------
    issue_halo(0);
    issue_w(0);
    issue_w(1);                                   // nkt >= 9
    issue_w(2);
    wait_vm_lgkm<2 * NWP>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    const int relu_in = g.relu_in;
    int kt = 0;
    for (int c = 0; c < nchunks; c++) {
        const char* halo = smem + (c & (NH - 1)) * HALO_BYTES;
        const bool more_halo = NH > 1 && c + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; tap++, kt++) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            // ======== load segment ========
            const char* wsl = smem + LDS_W + (kt & 3) * WSLOT;
            u32x4 af[TM][4], wf[TN][4];
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int hp = hp0[i] + dy * HALO_W + dx;
                const int a0 = hp * 128 + ((hi ^ ((hp >> 1) & 7)) << 4);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) af[i][ks] = *reinterpret_cast<const u32x4*>(halo + (a0 ^ (ks * 32)));
            }
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int ks = 0; ks < 4; ks++) wf[j][ks] = *reinterpret_cast<const u32x4*>(wsl + (w_off ^ (ks * 32)) + j * 4096);
            // DMA: next chunk's halo (first tap of a chunk; the other halo buffer was last read one barrier ago), weights 3 steps ahead
            const bool halo_now = tap == 0 && more_halo;
            if (halo_now) issue_halo(c + 1);
            const int ahead = nkt - 2 - kt;                    // how many of W(kt+2), W(kt+3) exist
            if (ahead >= 2) issue_w(kt + 3);
            const bool halo_pending = (tap <= 1) && more_halo; // halo pieces issued at tap 0 may still be in flight through tap 1
            if (halo_pending) {
                if (ahead >= 2) wait_vm_lgkm<2 * NWP + HPW>();
                else if (ahead == 1) wait_vm_lgkm<NWP + HPW>();
                else wait_vm_lgkm<HPW>();
            } else {
                if (ahead >= 2) wait_vm_lgkm<2 * NWP>();
                else if (ahead == 1) wait_vm_lgkm<NWP>();
                else wait_vm_lgkm<0>();
            }
            if (relu_in) {
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) af[i][ks] = relu8(af[i][ks]);
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ======== compute segment ========
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) mma_step<f16>(acc[i][j], wf[j][ks], af[i][ks]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            if (!(grp == 1 && kt == nkt - 1)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: bias / uv / ReLU in registers, transpose through LDS, (residual add,) 16-byte pixel-row stores -------------
    char* R = smem + wave * (WROWS * 128);
    const int rr = lane >> 3, cc = lane & 7;
    const int nw = n0 + wn * 64;
    float u[TM][2], vv[TM][2];
    if constexpr (HAS_UV) {
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int mp = wm * WROWS + i * 32 + l31;
            int y = y0 + (mp >> 4), x = x0 + (mp & 15);
            y = y < H ? y : H - 1; x = x < W ? x : W - 1;
            if constexpr (CONVT) {
#pragma unroll
                for (int d = 0; d < 2; d++) {     // high-res coordinates of the two output parities
                    u[i][d] = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, 2 * W, 2 * x + d);
                    vv[i][d] = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, 2 * H, 2 * y + d);
                }
            } else {
                u[i][0] = u[i][1] = linspace_at(g.uv.u0, g.uv.u1, g.uv.ustep, W, x);
                vv[i][0] = vv[i][1] = linspace_at(g.uv.v0, g.uv.v1, g.uv.vstep, H, y);
            }
        }
    }
    const float lo = g.act == ACT_RELU ? 0.f : -3.0e38f;        // branch-free optional ReLU
    const bool has_bias = g.bias != nullptr;
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
                    const float uu = pdx ? u[i][1] : u[i][0];
                    const float vq = pdy ? vv[i][1] : vv[i][0];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] += wu[e] * uu + wv[e] * vq;
                }
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], lo);
                const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4*>(R + row * 128 + ((((j * 4 + q) ^ (row & 7)) << 4) | (hi << 3))) = hv;
            }
        }
    f16* const outp = reinterpret_cast<f16*>(g.out);
    const f16* const addp = reinterpret_cast<const f16*>(g.add);
#pragma unroll
    for (int it = 0; it < WROWS / 8; it++) {
        const int row = it * 8 + rr;
        const int mp = wm * WROWS + row;
        const int y = y0 + (mp >> 4), x = x0 + (mp & 15);
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
                    const f16x8 a = *reinterpret_cast<const f16x8*>(addp + (((size_t)b * H + y) * W + x) * g.ldadd + nw + cc * 8);
                    f16x8 h = __builtin_bit_cast(f16x8, v);
                    h += a;                                  // fp16 + fp16, as the reference's .half() path does (x + conv(x))
                    v = __builtin_bit_cast(u32x4, h);
                }
                *reinterpret_cast<u32x4*>(outp + idx) = v;
            }
        }
    }
}

template <int BN, int NH, int EPI>
int launch_conv_cfg(const GemmArgs& g, hipStream_t st) {
    constexpr int smem = NH * HALO_BYTES + 4 * BN * 128;
    static bool attr_set = false;
    auto kern = conv_pp_kernel<BN, NH, EPI>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long B = (long)g.M / ((long)g.H * g.W);
    const long tiles = B * ((g.H + 15) / 16) * ((g.W + 15) / 16) * (g.N / BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), smem, st, g);
    return (int)hipGetLastError();
}

}  // namespace

// AMODE_CONV3 problems the halo kernel takes (f16): Cin multiple of 64, N = 64 or a multiple of 128, plain / pixel-shuffle store
bool conv_pp_eligible(const GemmArgs& g) {
    if ((g.C & 63) || g.K != 9 * g.C || g.ldw != 9 * g.C) return false;
    if (g.N != 64 && (g.N & 127)) return false;
    if (g.H < 1 || g.W < 1 || (long)g.M % ((long)g.H * g.W) != 0) return false;
    if ((long)g.H * g.W * g.C * 2 >= (1L << 31)) return false;                     // 32-bit halo offsets
    if (g.epi == EPI_STORE)        // (the residual add is applied after the activation here: never combined by the decoder)
        return (g.ldc & 7) == 0 && (!g.add || ((g.ldadd & 7) == 0 && g.act == ACT_NONE)) && (!g.uv.wu || g.bias) && g.act != ACT_GELU;
    if (g.epi == EPI_CONVT) return !g.add && g.act == ACT_NONE && (g.Cout == 32 || (g.Cout & 63) == 0) && g.N == 4 * g.Cout;
    return false;
}

template <int BN, int NH>
static int launch_conv_epi(const GemmArgs& g, hipStream_t st) {
    const int e = (g.uv.wu ? 1 : 0) | (g.epi == EPI_CONVT ? 2 : 0);
    switch (e) {
    case 0: return launch_conv_cfg<BN, NH, 0>(g, st);
    case 1: return launch_conv_cfg<BN, NH, 1>(g, st);
    case 2: return launch_conv_cfg<BN, NH, 2>(g, st);
    default: return launch_conv_cfg<BN, NH, 3>(g, st);
    }
}

int launch_conv_pp(const GemmArgs& g, hipStream_t st) {
    if (g.N == 64) return g.C == 64 ? launch_conv_epi<64, 1>(g, st) : launch_conv_epi<64, 2>(g, st);
    return g.C == 64 ? launch_conv_epi<128, 1>(g, st) : launch_conv_epi<128, 2>(g, st);
}
