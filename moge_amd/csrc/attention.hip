// Flash attention for the DINOv2 blocks (attention.py:70-81: softmax(q k^T / sqrt(64)) v, no mask, no dropout),
// head_dim = 64, N up to ~10k tokens, for gfx950.
//
// Inputs come from the QKV GEMM epilogue (gemm.hip, EPI_QKV): q,k as (B,nh,N,64) with q pre-multiplied by
// log2(e)/8, and v TRANSPOSED as vT (B,nh,64,Npad) so that both MFMAs contract over a memory-contiguous axis.
//
// Both products are issued "swapped" so the query index is the lane index:
//     S^T[key][query]  = K[key][d]   . Q^T[d][query]     (A-operand: K rows from LDS,  B-operand: Q rows in registers)
//     O^T[d][query]    = V^T[d][key] . P^T[key][query]   (A-operand: V^T rows from LDS, B-operand: P packed in registers)
// -> the online-softmax state (running max, running sum, rescale factor) is per LANE; the only cross-lane traffic
// is one exchange with lane^32 per KV tile.  The P accumulator registers of a lane feed the second MFMA's
// B-operand directly; for f16 that needs key(r) contiguous per lane half, obtained by loading K rows into the
// MFMA in a bit-2/bit-3 swapped order (perm32) - no data movement.
//
// Block = 4 waves x 32 queries; KV tile = 64 keys staged through LDS (16-byte chunks, XOR swizzle), next tile's
// global loads in flight during the MFMAs.  Output: (B, N, nh*64) storage type.
#include "common.h"

template <typename T> __device__ __forceinline__ int perm32(int i);
// f16: accumulator rows of a lane half are {0-3,8-11,..}; swapping bits 2/3 of the K row index makes regs
// [8s,8s+8) of lane-half hi hold keys 16s+8hi+(0..7) = exactly one 16-byte chunk of vT.
template <> __device__ __forceinline__ int perm32<f16>(int i) { return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1); }
// f32: regs [4s,4s+4) hold keys 8s+4hi+(0..3) already.
template <> __device__ __forceinline__ int perm32<float>(int i) { return i; }

template <typename T> __device__ __forceinline__ u32x4 pack_p(const f32x16& p, int s);
template <> __device__ __forceinline__ u32x4 pack_p<f16>(const f32x16& p, int s) {
    f16x8 h;
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = (f16)p[8 * s + i];
    return __builtin_bit_cast(u32x4, h);
}
template <> __device__ __forceinline__ u32x4 pack_p<float>(const f32x16& p, int s) {
    f32x4 h = {p[4 * s], p[4 * s + 1], p[4 * s + 2], p[4 * s + 3]};
    return __builtin_bit_cast(u32x4, h);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vT,
                                                   T* __restrict__ out, int Ntok, int Npad, int nh) {
    constexpr int CH = TT<T>::CH;
    constexpr int CPR = 64 / CH;                // chunks per 64-element tile row (8 or 16)
    constexpr int ROWB = 64 * (int)sizeof(T);   // tile row bytes
    constexpr int DSTEPS = 64 / (2 * CH);       // k-steps over head_dim (QK^T)
    constexpr int KSTEPS = 32 / (2 * CH);       // k-steps over a 32-key sub-tile (PV)
    constexpr int LD_IT = 64 * CPR / 256;       // chunks per thread per tile

    __shared__ __attribute__((aligned(16))) char sK[64 * ROWB];
    __shared__ __attribute__((aligned(16))) char sV[64 * ROWB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y;                  // b*nh + head
    const int b = bh / nh, head = bh - b * nh;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int qrow = q0 + l31;
    const int qld = qrow < Ntok ? qrow : Ntok - 1;

    const T* qp = q + ((size_t)bh * Ntok + qld) * 64;
    u32x4 qf[DSTEPS];
#pragma unroll
    for (int s = 0; s < DSTEPS; s++) qf[s] = *reinterpret_cast<const u32x4*>(qp + (2 * s + hi) * CH);

    const T* kbase = k + (size_t)bh * Ntok * 64;
    const T* vbase = vT + (size_t)bh * 64 * Npad;

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int ntiles = (Ntok + 63) / 64;
    u32x4 rk[LD_IT], rv[LD_IT];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < LD_IT; i++) {
            const int cq = tid + i * 256;
            const int row = cq / CPR, c = cq - row * CPR;
            int key = t * 64 + row;
            key = key < Ntok ? key : Ntok - 1;                      // clamped rows are masked below
            rk[i] = *reinterpret_cast<const u32x4*>(kbase + (size_t)key * 64 + c * CH);
            rv[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)row * Npad + t * 64 + c * CH);   // pad columns are zero
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < LD_IT; i++) {
            const int cq = tid + i * 256;
            const int row = cq / CPR, c = cq - row * CPR;
            *reinterpret_cast<u32x4*>(sK + row * ROWB + (swz<CPR>(row, c) << 4)) = rk[i];
            *reinterpret_cast<u32x4*>(sV + row * ROWB + (swz<CPR>(row, c) << 4)) = rv[i];
        }
    };

    load_tile(0);
    for (int t = 0; t < ntiles; t++) {
        __syncthreads();                 // everyone finished reading the previous tile
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile(t + 1);

        // ---- S^T = K . Q^T for two 32-key sub-tiles ------------------------------------------------
        f32x16 sc[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int r = 0; r < 16; r++) sc[h][r] = 0.f;
            const int row = h * 32 + perm32<T>(l31);
#pragma unroll
            for (int s = 0; s < DSTEPS; s++) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(sK + row * ROWB + (swz<CPR>(row, 2 * s + hi) << 4));
                mma_step<T>(sc[h], kf, qf[s]);
            }
        }
        // register r of sub-tile h holds key  t*64 + h*32 + perm32(acc_row(r,hi))
        if (t == ntiles - 1) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = t * 64 + h * 32 + perm32<T>(acc_row(r, hi));
                    if (key >= Ntok) sc[h][r] = -1e30f;
                }
        }
        // ---- online softmax (scores are already in log2 units) -------------------------------------
        float mx = sc[0][0];
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, sc[h][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float p = exp2f(sc[h][r] - m_new);
                sc[h][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[i][r] *= alpha;
        // ---- O^T += V^T . P^T ------------------------------------------------------------------------
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int s = 0; s < KSTEPS; s++) {
                const u32x4 pf = pack_p<T>(sc[h], s);
                const int c = h * (32 / CH) + 2 * s + hi;
#pragma unroll
                for (int dt = 0; dt < 2; dt++) {
                    const int row = dt * 32 + l31;
                    const u32x4 vf = *reinterpret_cast<const u32x4*>(sV + row * ROWB + (swz<CPR>(row, c) << 4));
                    mma_step<T>(o[dt], vf, pf);
                }
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    if (qrow < Ntok) {
        T* op = out + ((size_t)b * Ntok + qrow) * ((size_t)nh * 64) + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++)
                store4(op + dt * 32 + 8 * g4 + 4 * hi, o[dt][4 * g4] * inv, o[dt][4 * g4 + 1] * inv, o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv);
    }
}

template <typename T>
int launch_attention(const void* q, const void* k, const void* vT, void* out, int B, int nh, int Ntok, int Npad, hipStream_t st) {
    dim3 grid((Ntok + 127) / 128, B * nh);
    hipLaunchKernelGGL(attn_kernel<T>, grid, dim3(256), 0, st, (const T*)q, (const T*)k, (const T*)vT, (T*)out, Ntok, Npad, nh);
    return (int)hipGetLastError();
}
template int launch_attention<f16>(const void*, const void*, const void*, void*, int, int, int, int, hipStream_t);
template int launch_attention<float>(const void*, const void*, const void*, void*, int, int, int, int, hipStream_t);
