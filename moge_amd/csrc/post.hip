// Decoder tail + affine-invariant point-map recovery for gfx950.
//   head_final : 1x1 output conv (modules.py:231) + bilinear resize to the image size (v2.py:170) + remap / normalise /
//                sigmoid (v2.py:173-180), fused - the (B,3,16h0,16w0) head output is never materialised
//   mlp_layer  : scale head (modules.py:184-192, v2.py:167,182)
//   recover    : recover_focal_shift (geometry_torch.py:115-170) = nearest 64x64 subsample + per-image MINPACK lmdif
//                (scipy least_squares(method='lm'), geometry_numpy.py:79-112) as ONE workgroup per image: fp64
//                wavefront reductions for every residual evaluation, scalar LM state machine replicated in all lanes;
//                also writes the intrinsics (v2.py:265-266).  No device->host round trip.
//   finalize   : z += shift, mask &= z>0, depth, re-projection, metric scale, masking (v2.py:267-289), one pass.
#include "common.h"
#include "../../include/moge_hip.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------------------------
template <typename T, int KIND>   // KIND 0 points, 1 normal, 2 mask (sigmoid), 3 raw single channel (MoGe-1 mask, v1.py:289: no activation)
__global__ __launch_bounds__(256) void head_final_kernel(const T* __restrict__ x4, const float* __restrict__ w, const float* __restrict__ bias,
                                                         const T* __restrict__ n4, const float* __restrict__ w2,
                                                         float* __restrict__ out, int B, int Hd, int Wd, int C, int ld, int H, int W, int remap) {
    // ld = channel pitch of x4 / n4 (>= C: x4 may point at a channel slice of a wider map)
    // optional second input n4 (same shape as x4) with its own 1x1 weights w2: the level-4 input block of the head
    // (x + in_4(neck_4), modules.py:245) pre-composed with the output conv - both are linear and commute with the bilinear
    // resize, so the 32-channel sum never has to be materialised at 16x the token resolution
    constexpr int CH = TT<T>::CH;
    constexpr int CO = (KIND == 2 || KIND == 3) ? 1 : 3;
    __shared__ float sw[2 * 3 * 64];
    for (int i = threadIdx.x; i < CO * C; i += 256) { sw[i] = w[i]; sw[CO * C + i] = n4 ? w2[i] : 0.f; }
    __syncthreads();
    const long total = (long)B * H * W;
    const float sy_scale = (float)Hd / (float)H, sx_scale = (float)Wd / (float)W;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ox = idx % W;
        const long t = idx / W;
        const int oy = t % H, b = t / H;
        float sy = sy_scale * (oy + 0.5f) - 0.5f, sx = sx_scale * (ox + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 < Hd - 1 ? y0 : Hd - 1; x0 = x0 < Wd - 1 ? x0 : Wd - 1;
        const int y1 = y0 + 1 < Hd ? y0 + 1 : Hd - 1, x1 = x0 + 1 < Wd ? x0 + 1 : Wd - 1;
        float ly = sy - y0, lx = sx - x0;
        ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        const T* base = x4 + (size_t)b * Hd * Wd * ld;
        const T* p00 = base + ((size_t)y0 * Wd + x0) * ld;
        const T* p01 = base + ((size_t)y0 * Wd + x1) * ld;
        const T* p10 = base + ((size_t)y1 * Wd + x0) * ld;
        const T* p11 = base + ((size_t)y1 * Wd + x1) * ld;
        float o[CO];
#pragma unroll
        for (int j = 0; j < CO; j++) o[j] = bias[j];
        for (int c0 = 0; c0 < C; c0 += CH) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(p00 + c0), bq = *reinterpret_cast<const u32x4*>(p01 + c0);
            const u32x4 cq = *reinterpret_cast<const u32x4*>(p10 + c0), dq = *reinterpret_cast<const u32x4*>(p11 + c0);
            float f[CH];
            if constexpr (CH == 8) {
                const f16x8 va = __builtin_bit_cast(f16x8, a), vb = __builtin_bit_cast(f16x8, bq), vc = __builtin_bit_cast(f16x8, cq), vd = __builtin_bit_cast(f16x8, dq);
#pragma unroll
                for (int i = 0; i < CH; i++) f[i] = w00 * (float)va[i] + w01 * (float)vb[i] + w10 * (float)vc[i] + w11 * (float)vd[i];
            } else {
                const f32x4 va = __builtin_bit_cast(f32x4, a), vb = __builtin_bit_cast(f32x4, bq), vc = __builtin_bit_cast(f32x4, cq), vd = __builtin_bit_cast(f32x4, dq);
#pragma unroll
                for (int i = 0; i < CH; i++) f[i] = w00 * va[i] + w01 * vb[i] + w10 * vc[i] + w11 * vd[i];
            }
#pragma unroll
            for (int j = 0; j < CO; j++)
#pragma unroll
                for (int i = 0; i < CH; i++) o[j] += sw[j * C + c0 + i] * f[i];
        }
        if (n4) {
            const size_t d = n4 - x4;            // same indexing in the second tensor
            for (int c0 = 0; c0 < C; c0 += CH) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(p00 + d + c0), bq = *reinterpret_cast<const u32x4*>(p01 + d + c0);
                const u32x4 cq = *reinterpret_cast<const u32x4*>(p10 + d + c0), dq = *reinterpret_cast<const u32x4*>(p11 + d + c0);
                float f[CH];
                if constexpr (CH == 8) {
                    const f16x8 va = __builtin_bit_cast(f16x8, a), vb = __builtin_bit_cast(f16x8, bq), vc = __builtin_bit_cast(f16x8, cq), vd = __builtin_bit_cast(f16x8, dq);
#pragma unroll
                    for (int i = 0; i < CH; i++) f[i] = w00 * (float)va[i] + w01 * (float)vb[i] + w10 * (float)vc[i] + w11 * (float)vd[i];
                } else {
                    const f32x4 va = __builtin_bit_cast(f32x4, a), vb = __builtin_bit_cast(f32x4, bq), vc = __builtin_bit_cast(f32x4, cq), vd = __builtin_bit_cast(f32x4, dq);
#pragma unroll
                    for (int i = 0; i < CH; i++) f[i] = w00 * va[i] + w01 * vb[i] + w10 * vc[i] + w11 * vd[i];
                }
#pragma unroll
                for (int j = 0; j < CO; j++)
#pragma unroll
                    for (int i = 0; i < CH; i++) o[j] += sw[CO * C + j * C + c0 + i] * f[i];
            }
        }
        if (KIND == 0) {
            float x = o[0], y = o[1], z = o[2];
            if (remap == MOGE_REMAP_EXP) { z = expf(z); x *= z; y *= z; }
            else if (remap == MOGE_REMAP_SINH) { x = sinhf(x); y = sinhf(y); z = sinhf(z); }
            else if (remap == MOGE_REMAP_SINH_EXP) { x = sinhf(x); y = sinhf(y); z = expf(z); }
            out[idx * 3] = x; out[idx * 3 + 1] = y; out[idx * 3 + 2] = z;
        } else if (KIND == 1) {
            const float nrm = fmaxf(sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]), 1e-12f);
            out[idx * 3] = o[0] / nrm; out[idx * 3 + 1] = o[1] / nrm; out[idx * 3 + 2] = o[2] / nrm;
        } else if (KIND == 3) {
            out[idx] = o[0];
        } else {
            out[idx] = 1.f / (1.f + expf(-o[0]));
        }
    }
}
// fp16, C == 32 fast path: FOUR lanes per output pixel (8 channels = one 16-byte load per lane and tap, a lane quad reads a
// pixel's 64 contiguous bytes), partial 1x1 products reduced over the quad with two DPP-class shuffles.
template <int KIND>
__global__ __launch_bounds__(256) void head_final32_kernel(const f16* __restrict__ x4, const float* __restrict__ w, const float* __restrict__ bias,
                                                           const f16* __restrict__ n4, const float* __restrict__ w2,
                                                           float* __restrict__ out, int B, int Hd, int Wd, int ld, int H, int W, int remap) {
    constexpr int C = 32, CO = (KIND == 2 || KIND == 3) ? 1 : 3;
    const int sub = threadIdx.x & 3;
    float wr[CO][8], wr2[CO][8];
#pragma unroll
    for (int j = 0; j < CO; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) { wr[j][i] = w[j * C + sub * 8 + i]; wr2[j][i] = n4 ? w2[j * C + sub * 8 + i] : 0.f; }
    const long total = (long)B * H * W;
    const float sy_scale = (float)Hd / (float)H, sx_scale = (float)Wd / (float)W;
    for (long idx = (blockIdx.x * 256L + threadIdx.x) >> 2; idx < total; idx += (long)gridDim.x * 64) {
        const int ox = idx % W;
        const long t = idx / W;
        const int oy = t % H, b = t / H;
        float sy = sy_scale * (oy + 0.5f) - 0.5f, sx = sx_scale * (ox + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 < Hd - 1 ? y0 : Hd - 1; x0 = x0 < Wd - 1 ? x0 : Wd - 1;
        const int y1 = y0 + 1 < Hd ? y0 + 1 : Hd - 1, x1 = x0 + 1 < Wd ? x0 + 1 : Wd - 1;
        float ly = sy - y0, lx = sx - x0;
        ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
        const float wt[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
        const size_t base = (size_t)b * Hd * Wd * ld + sub * 8;
        const size_t off[4] = {base + ((size_t)y0 * Wd + x0) * ld, base + ((size_t)y0 * Wd + x1) * ld, base + ((size_t)y1 * Wd + x0) * ld,
                               base + ((size_t)y1 * Wd + x1) * ld};
        u32x4 qa[4], qb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) qa[k] = *reinterpret_cast<const u32x4*>(x4 + off[k]);
        if (n4) {
#pragma unroll
            for (int k = 0; k < 4; k++) qb[k] = *reinterpret_cast<const u32x4*>(n4 + off[k]);
        }
        float o[CO];
#pragma unroll
        for (int j = 0; j < CO; j++) o[j] = 0.f;
        {
            float f[8];
#pragma unroll
            for (int i = 0; i < 8; i++) f[i] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f16x8 v = __builtin_bit_cast(f16x8, qa[k]);
#pragma unroll
                for (int i = 0; i < 8; i++) f[i] += wt[k] * (float)v[i];
            }
#pragma unroll
            for (int j = 0; j < CO; j++)
#pragma unroll
                for (int i = 0; i < 8; i++) o[j] += wr[j][i] * f[i];
        }
        if (n4) {
            float f[8];
#pragma unroll
            for (int i = 0; i < 8; i++) f[i] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f16x8 v = __builtin_bit_cast(f16x8, qb[k]);
#pragma unroll
                for (int i = 0; i < 8; i++) f[i] += wt[k] * (float)v[i];
            }
#pragma unroll
            for (int j = 0; j < CO; j++)
#pragma unroll
                for (int i = 0; i < 8; i++) o[j] += wr2[j][i] * f[i];
        }
#pragma unroll
        for (int j = 0; j < CO; j++) {
            o[j] += __shfl_xor(o[j], 1);
            o[j] += __shfl_xor(o[j], 2);
            o[j] += bias[j];
        }
        if (sub != 0) continue;
        if (KIND == 0) {
            float x = o[0], y = o[1], z = o[2];
            if (remap == MOGE_REMAP_EXP) { z = expf(z); x *= z; y *= z; }
            else if (remap == MOGE_REMAP_SINH) { x = sinhf(x); y = sinhf(y); z = sinhf(z); }
            else if (remap == MOGE_REMAP_SINH_EXP) { x = sinhf(x); y = sinhf(y); z = expf(z); }
            out[idx * 3] = x; out[idx * 3 + 1] = y; out[idx * 3 + 2] = z;
        } else if (KIND == 1) {
            const float nrm = fmaxf(sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]), 1e-12f);
            out[idx * 3] = o[0] / nrm; out[idx * 3 + 1] = o[1] / nrm; out[idx * 3 + 2] = o[2] / nrm;
        } else if (KIND == 3) {
            out[idx] = o[0];
        } else {
            out[idx] = 1.f / (1.f + expf(-o[0]));
        }
    }
}

template <typename T>
int launch_head_final(int kind, const void* x4, const float* w, const float* bias, const void* n4, const float* w2, float* out, int B, int Hd,
                      int Wd, int C, int H, int W, int remap, hipStream_t st, int ld) {
    if (ld <= 0) ld = C;
    if (C > 64 || C % TT<T>::CH != 0 || ld % TT<T>::CH != 0) return -1;
    const long total = (long)B * H * W;
    if (std::is_same<T, f16>::value && C == 32) {
        long nb = (total * 4 + 255) / 256;
        const int blocks4 = (int)(nb > 65536 ? 65536 : nb);
        const f16* xa = (const f16*)x4; const f16* xb = (const f16*)n4;
#define HF32(K) hipLaunchKernelGGL((head_final32_kernel<K>), dim3(blocks4), dim3(256), 0, st, xa, w, bias, xb, w2, out, B, Hd, Wd, ld, H, W, remap)
        if (kind == 0) HF32(0); else if (kind == 1) HF32(1); else if (kind == 2) HF32(2); else HF32(3);
#undef HF32
        return (int)hipGetLastError();
    }
    int blocks = (int)((total + 255) / 256);
    if (blocks > 32768) blocks = 32768;
#define HFG(K) hipLaunchKernelGGL((head_final_kernel<T, K>), dim3(blocks), dim3(256), 0, st, (const T*)x4, w, bias, (const T*)n4, w2, out, B, Hd, Wd, C, ld, H, W, remap)
    if (kind == 0) HFG(0); else if (kind == 1) HFG(1); else if (kind == 2) HFG(2); else HFG(3);
#undef HFG
    return (int)hipGetLastError();
}
template int launch_head_final<f16>(int, const void*, const float*, const float*, const void*, const float*, float*, int, int, int, int, int, int, int, hipStream_t, int);
template int launch_head_final<float>(int, const void*, const float*, const float*, const void*, const float*, float*, int, int, int, int, int, int, int, hipStream_t, int);

// head_final with a 3x3 last conv (MoGe-1 `last_conv_size` 3, v1.py:108: replicate padding): out = remap(resize(conv3x3(x) + bias)).  Resize and conv
// are both linear: every output pixel takes its four bilinear taps of the conv evaluated on the fly (4 x 9 positions x C channels, fp32 weights in
// LDS in the checkpoint's own [CO][C][3][3] layout) - no intermediate map, no rounding between the conv and the resize.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void head_final_k3_kernel(const T* __restrict__ x4, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                                                            int B, int Hd, int Wd, int C, int H, int W, int remap) {
    constexpr int CH = TT<T>::CH;
    constexpr int CO = (KIND == 2 || KIND == 3) ? 1 : 3;
    __shared__ float sw[3 * 64 * 9];
    for (int i = threadIdx.x; i < CO * C * 9; i += 256) sw[i] = w[i];
    __syncthreads();
    const long total = (long)B * H * W;
    const float sy_scale = (float)Hd / (float)H, sx_scale = (float)Wd / (float)W;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ox = idx % W;
        const long t = idx / W;
        const int oy = t % H, b = t / H;
        float sy = sy_scale * (oy + 0.5f) - 0.5f, sx = sx_scale * (ox + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 < Hd - 1 ? y0 : Hd - 1; x0 = x0 < Wd - 1 ? x0 : Wd - 1;
        const int y1 = y0 + 1 < Hd ? y0 + 1 : Hd - 1, x1 = x0 + 1 < Wd ? x0 + 1 : Wd - 1;
        float ly = sy - y0, lx = sx - x0;
        ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
        const float wt[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
        const int ty[4] = {y0, y0, y1, y1}, tx[4] = {x0, x1, x0, x1};
        const T* base = x4 + (size_t)b * Hd * Wd * C;
        float o[CO];
#pragma unroll
        for (int j = 0; j < CO; j++) o[j] = bias[j];
        for (int k = 0; k < 4; k++) {
            if (wt[k] == 0.f) continue;
            float acc[CO];
#pragma unroll
            for (int j = 0; j < CO; j++) acc[j] = 0.f;
            for (int dy = -1; dy <= 1; dy++) {
                int py = ty[k] + dy; py = py < 0 ? 0 : (py > Hd - 1 ? Hd - 1 : py);
                for (int dx = -1; dx <= 1; dx++) {
                    int px = tx[k] + dx; px = px < 0 ? 0 : (px > Wd - 1 ? Wd - 1 : px);
                    const T* p = base + ((size_t)py * Wd + px) * C;
                    const int tap = (dy + 1) * 3 + dx + 1;
                    for (int c0 = 0; c0 < C; c0 += CH) {
                        float f[CH];
                        const u32x4 a = *reinterpret_cast<const u32x4*>(p + c0);
                        if constexpr (CH == 8) {
                            const f16x8 va = __builtin_bit_cast(f16x8, a);
#pragma unroll
                            for (int i = 0; i < CH; i++) f[i] = (float)va[i];
                        } else {
                            const f32x4 va = __builtin_bit_cast(f32x4, a);
#pragma unroll
                            for (int i = 0; i < CH; i++) f[i] = va[i];
                        }
#pragma unroll
                        for (int j = 0; j < CO; j++)
#pragma unroll
                            for (int i = 0; i < CH; i++) acc[j] = fmaf(sw[(j * C + c0 + i) * 9 + tap], f[i], acc[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < CO; j++) o[j] = fmaf(wt[k], acc[j], o[j]);
        }
        if (KIND == 0) {
            float x = o[0], y = o[1], z = o[2];
            if (remap == MOGE_REMAP_EXP) { z = expf(z); x *= z; y *= z; }
            else if (remap == MOGE_REMAP_SINH) { x = sinhf(x); y = sinhf(y); z = sinhf(z); }
            else if (remap == MOGE_REMAP_SINH_EXP) { x = sinhf(x); y = sinhf(y); z = expf(z); }
            out[idx * 3] = x; out[idx * 3 + 1] = y; out[idx * 3 + 2] = z;
        } else {
            out[idx] = o[0];
        }
    }
}
// kind 0: points (3 channels + remap), kind 3: raw mask (1 channel); x (B,Hd,Wd,C) dense, C <= 64
template <typename T>
int launch_head_final_k3(int kind, const void* x4, const float* w, const float* bias, float* out, int B, int Hd, int Wd, int C, int H, int W, int remap, hipStream_t st) {
    if (C > 64 || C % TT<T>::CH != 0 || (kind != 0 && kind != 3)) return -1;
    const long total = (long)B * H * W;
    long nb = (total + 255) / 256;
    const int blocks = (int)(nb > 65536 ? 65536 : nb);
    if (kind == 0) hipLaunchKernelGGL((head_final_k3_kernel<T, 0>), dim3(blocks), dim3(256), 0, st, (const T*)x4, w, bias, out, B, Hd, Wd, C, H, W, remap);
    else hipLaunchKernelGGL((head_final_k3_kernel<T, 3>), dim3(blocks), dim3(256), 0, st, (const T*)x4, w, bias, out, B, Hd, Wd, C, H, W, remap);
    return (int)hipGetLastError();
}
template int launch_head_final_k3<f16>(int, const void*, const float*, const float*, float*, int, int, int, int, int, int, int, hipStream_t);
template int launch_head_final_k3<float>(int, const void*, const float*, const float*, float*, int, int, int, int, int, int, int, hipStream_t);

// head_final on the maps of the fused output conv (conv_pp.hip, EPI bit 5): the 1x1 output conv (modules.py:231) and the pre-composed level-4
// input block (modules.py:245) were applied per high-res pixel by the resampler kernels, so what is left is 4 + 4 floats per tap:
// bilinear resize (v2.py:170; linear: it commutes with the convs), bias, remap (v2.py:173-180).  One thread per output pixel.
template <int KIND>
__global__ __launch_bounds__(256) void head_final_dot_kernel(const float* __restrict__ y, const float* __restrict__ z, int zld, int zoff,
                                                             const float* __restrict__ bias, float* __restrict__ out, int B, int Hd, int Wd, int H, int W, int remap) {
    constexpr int CO = (KIND == 2 || KIND == 3) ? 1 : 3;
    const long total = (long)B * H * W;
    const float sy_scale = (float)Hd / (float)H, sx_scale = (float)Wd / (float)W;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ox = idx % W;
        const long t = idx / W;
        const int oy = t % H, b = t / H;
        float sy = sy_scale * (oy + 0.5f) - 0.5f, sx = sx_scale * (ox + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 < Hd - 1 ? y0 : Hd - 1; x0 = x0 < Wd - 1 ? x0 : Wd - 1;
        const int y1 = y0 + 1 < Hd ? y0 + 1 : Hd - 1, x1 = x0 + 1 < Wd ? x0 + 1 : Wd - 1;
        float ly = sy - y0, lx = sx - x0;
        ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
        const float wt[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
        const size_t pb = (size_t)b * Hd * Wd;
        const size_t pix[4] = {pb + (size_t)y0 * Wd + x0, pb + (size_t)y0 * Wd + x1, pb + (size_t)y1 * Wd + x0, pb + (size_t)y1 * Wd + x1};
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            f32x4 v = *reinterpret_cast<const f32x4*>(y + pix[k] * 4);
            if (z) v += *reinterpret_cast<const f32x4*>(z + pix[k] * zld + zoff);
            acc += wt[k] * v;
        }
        float o[3];
#pragma unroll
        for (int j = 0; j < CO; j++) o[j] = acc[j] + bias[j];
        if (KIND == 0) {
            float x = o[0], yy = o[1], zz = o[2];
            if (remap == MOGE_REMAP_EXP) { zz = expf(zz); x *= zz; yy *= zz; }
            else if (remap == MOGE_REMAP_SINH) { x = sinhf(x); yy = sinhf(yy); zz = sinhf(zz); }
            else if (remap == MOGE_REMAP_SINH_EXP) { x = sinhf(x); yy = sinhf(yy); zz = expf(zz); }
            out[idx * 3] = x; out[idx * 3 + 1] = yy; out[idx * 3 + 2] = zz;
        } else if (KIND == 1) {
            const float nrm = fmaxf(sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]), 1e-12f);
            out[idx * 3] = o[0] / nrm; out[idx * 3 + 1] = o[1] / nrm; out[idx * 3 + 2] = o[2] / nrm;
        } else if (KIND == 3) {
            out[idx] = o[0];
        } else {
            out[idx] = 1.f / (1.f + expf(-o[0]));
        }
    }
}
int launch_head_final_dot(int kind, const float* y, const float* z, int zld, int zoff, const float* bias, float* out, int B, int Hd, int Wd, int H, int W,
                          int remap, hipStream_t st) {
    const long total = (long)B * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65536) blocks = 65536;
#define HFD(K) hipLaunchKernelGGL((head_final_dot_kernel<K>), dim3(blocks), dim3(256), 0, st, y, z, zld, zoff, bias, out, B, Hd, Wd, H, W, remap)
    if (kind == 0) HFD(0); else if (kind == 1) HFD(1); else if (kind == 2) HFD(2); else HFD(3);
#undef HFD
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// out[b][n] = f(sum_k in[b][k]*W[n][k] + bias[n]); one wave per output, fp32.  act: 0 none, 1 relu, 2 exp
__global__ __launch_bounds__(256) void mlp_layer_kernel(const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
                                                        float* __restrict__ out, int B, int K, int N, int act) {
    const int lane = threadIdx.x & 63;
    const long o = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (o >= (long)B * N) return;
    const int b = o / N, n = o - (long)b * N;
    const float* x = in + (size_t)b * K;
    const float* wr = W + (size_t)n * K;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + k), w4 = *reinterpret_cast<const f32x4*>(wr + k);
        s += a[0] * w4[0] + a[1] * w4[1] + a[2] * w4[2] + a[3] * w4[3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        s += bias[n];
        if (act == 1) s = fmaxf(s, 0.f);
        else if (act == 2) s = expf(s);
        out[o] = s;
    }
}
int launch_mlp_layer(const float* in, const float* W, const float* bias, float* out, int B, int K, int N, int act, hipStream_t st) {
    if (K % 4 != 0) return -1;
    hipLaunchKernelGGL(mlp_layer_kernel, dim3((unsigned)(((long)B * N + 3) / 4)), dim3(256), 0, st, in, W, bias, out, B, K, N, act);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// recovery
// ------------------------------------------------------------------------------------------------------------
// REC_THREADS threads x REC_PTS sample points = the 64 x 64 grid.  Batch-1 latency: the solver is a chain of ~35 block reductions over fp64 divisions; more
// waves shorten every link (round 3: 4 waves 168 us; 16 waves 82-91 us with 300 B per lane of scratch under its 128-register budget; 8 waves, no scratch: 89-90 us -
// level then).  Round 6, with the residuals / Jacobian column of the finite-difference pass kept per lane for the Householder pass: 16 waves 67.7 us (those arrays
// live in scratch under 128 registers), **8 waves 30.8 us** (256 registers, no scratch), 4 waves 41.1 us (profiles/r06ae_recover_threads.log) - the default is 512
// threads from round 6 on; the summation tree of the block reductions differs from the 16-wave form's in the last fp64 bits only (same gates, same fixtures).

template <int NV, int REC_THREADS>
__device__ __forceinline__ void block_sum(double* v, double* sh) {
    constexpr int NW = REC_THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        double x = v[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0) sh[wave * NV + j] = x;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; j++) {
        double t[4];                       // fixed tree: four chains of NW / 4 waves, then ((t0 + t1) + t2) + t3
#pragma unroll
        for (int q = 0; q < 4; q++) {
            t[q] = sh[(q * (NW / 4)) * NV + j];
#pragma unroll
            for (int w = 1; w < NW / 4; w++) t[q] += sh[(q * (NW / 4) + w) * NV + j];
        }
        v[j] = ((t[0] + t[1]) + t[2]) + t[3];
    }
    __syncthreads();
}

template <int REC_PTS>
struct RecPts {
    float x[REC_PTS], y[REC_PTS], z[REC_PTS], u[REC_PTS], v[REC_PTS];
    unsigned valid;     // bit i: sample i of this thread is inside the mask
};

// closed-form focal for a given shift (geometry_numpy.py:85-86), fp64
template <int REC_PTS>
__device__ __forceinline__ void focal_sums(const RecPts<REC_PTS>& P, double s, double& num, double& den) {
    num = 0.0; den = 0.0;
#pragma unroll
    for (int i = 0; i < REC_PTS; i++)
        if (P.valid >> i & 1) {
            const double d = (double)P.z[i] + s;
            const double px = (double)P.x[i] / d, py = (double)P.y[i] / d;
            num += px * (double)P.u[i] + py * (double)P.v[i];
            den += px * px + py * py;
        }
}

struct LmState { double par; };

// MINPACK lmpar specialised to n = 1 (oracle/lmdif.py::_lmpar1)
__device__ void lmpar1(double r, double diag, double qtb, double delta, double& par, double& x) {
    const double DWARF = 2.2250738585072014e-308;
    x = r != 0.0 ? qtb / r : 0.0;
    const bool nonsing = r != 0.0;
    int it = 0;
    double dxnorm = fabs(diag * x);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { par = 0.0; return; }
    double parl = 0.0;
    if (nonsing) {
        const double w = diag * (diag * x / dxnorm) / r;
        const double temp = fabs(w);
        parl = ((fp / delta) / temp) / temp;
    }
    const double gnorm = fabs(r * qtb / diag);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = DWARF / fmin(delta, 0.1);
    par = fmax(par, parl);
    par = fmin(par, paru);
    if (par == 0.0) par = gnorm / dxnorm;
    for (;;) {
        it++;
        if (par == 0.0) par = fmax(DWARF, 0.001 * paru);
        const double d = sqrt(par) * diag;
        double sdiag, wa;
        if (d == 0.0) { sdiag = r; wa = qtb; }
        else {
            double sn, cs;
            if (fabs(r) < fabs(d)) { const double cot = r / d; sn = 0.5 / sqrt(0.25 + 0.25 * cot * cot); cs = sn * cot; }
            else { const double tn = d / r; cs = 0.5 / sqrt(0.25 + 0.25 * tn * tn); sn = cs * tn; }
            sdiag = cs * r + sn * d;
            wa = cs * qtb;
        }
        x = sdiag != 0.0 ? wa / sdiag : 0.0;
        dxnorm = fabs(diag * x);
        const double temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || it == 10) break;
        const double w = diag * (diag * x / dxnorm) / sdiag;
        const double t = fabs(w);
        const double parc = ((fp / delta) / t) / t;
        if (fp > 0.0) parl = fmax(parl, par);
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
}

// One workgroup per image.
//   points (B,H,W,3) fp32; validity from mask_prob (>0.5f) or mask_u8 (!=0) or all-valid if both null
//   fov_deg: null -> solve focal and shift; else focal fixed from fov_x (v2.py:261-263)
//   focal_in: optional explicit focal (test entry point)
template <int REC_THREADS>
__global__ __launch_bounds__(REC_THREADS) void recover_kernel(const float* __restrict__ points, const float* __restrict__ mask_prob,
                                                      const uint8_t* __restrict__ mask_u8, const float* __restrict__ fov_deg,
                                                      const float* __restrict__ focal_in, int H, int W,
                                                      float u0, float u1, float ustep, float v0, float v1, float vstep,
                                                      float fov_c, float fx_mul, float fx_div, float fy_mul, float mask_thr,
                                                      float* __restrict__ focal_out, float* __restrict__ shift_out,
                                                      float* __restrict__ intrinsics, int* __restrict__ status) {
    constexpr int NW = REC_THREADS / 64, REC_PTS = 4096 / REC_THREADS;
    __shared__ double sh[NW * 2];
    __shared__ double sh0[2];
    __shared__ int shi[NW];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* pts = points + (size_t)b * H * W * 3;
    RecPts<REC_PTS> P;
    P.valid = 0;
    int first = 1 << 30, count = 0;
#pragma unroll
    for (int i = 0; i < REC_PTS; i++) {
        const int s = tid + REC_THREADS * i;           // raster index in the 64x64 grid
        const int gy = s >> 6, gx = s & 63;
        const int sy = (int)(((long)gy * H) >> 6), sx = (int)(((long)gx * W) >> 6);   // nearest: floor(dst*in/64)
        const size_t pix = (size_t)sy * W + sx;
        P.x[i] = pts[pix * 3]; P.y[i] = pts[pix * 3 + 1]; P.z[i] = pts[pix * 3 + 2];
        P.u[i] = linspace_at(u0, u1, ustep, W, sx);
        P.v[i] = linspace_at(v0, v1, vstep, H, sy);
        bool ok = true;
        if (mask_prob) ok = mask_prob[(size_t)b * H * W + pix] > mask_thr;
        else if (mask_u8) ok = mask_u8[(size_t)b * H * W + pix] != 0;
        if (ok) { P.valid |= 1u << i; count++; if (s < first) first = s; }
    }
    // block reduce count / first valid raster index
    for (int off = 32; off > 0; off >>= 1) { count += __shfl_xor(count, off); const int o = __shfl_xor(first, off); first = o < first ? o : first; }
    if ((tid & 63) == 0) { shi[tid >> 6] = count; }
    __syncthreads();
    count = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) count += shi[w];
    __syncthreads();
    if ((tid & 63) == 0) shi[tid >> 6] = first;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; w++) first = min(first, shi[w]);
    __syncthreads();
    const bool fixed = (fov_deg != nullptr) || (focal_in != nullptr);
    float focal_f = 1.f;
    if (focal_in) focal_f = focal_in[b];
    else if (fov_deg) focal_f = fov_c / tanf((fov_deg[b] / 2.f) * 0.017453292519943295f);
    const double focal_fixed = (double)focal_f;
    const bool own_first = (first % REC_THREADS) == tid;
    const int first_slot = first / REC_THREADS;

    float shift_res = 0.f, focal_res = fixed ? focal_f : 1.f;
    int st = 0;
    if (count >= 2) {
        const double EPSMCH = 2.220446049250313e-16;
        const double ftol = 1e-3, xtol = 1e-8, gtol = 1e-8, factor = 100.0, diag = 1.0;
        const int maxfev = 200;
        const double eps = sqrt(EPSMCH);
        double x = 0.0, par = 0.0, delta = 0.0, xnorm = 0.0;
        int iter = 1, info = 0, nfev = 1;
        // f(x0): focal + |fvec|
        double red[8];
        double f_cur = focal_fixed;
        if (!fixed) { focal_sums(P, x, red[0], red[1]); block_sum<2, REC_THREADS>(red, sh); f_cur = red[0] / red[1]; }
        red[0] = 0.0; red[1] = 0.0;
#pragma unroll
        for (int i = 0; i < REC_PTS; i++)
            if (P.valid >> i & 1) {
                const double d = (double)P.z[i] + x;
                const double rx = f_cur * ((double)P.x[i] / d) - (double)P.u[i], ry = f_cur * ((double)P.y[i] / d) - (double)P.v[i];
                red[0] += rx * rx + ry * ry;
                if (!(isfinite(rx) && isfinite(ry))) red[1] += 1.0;
            }
        block_sum<2, REC_THREADS>(red, sh);
        double fnorm = sqrt(red[0]);
        if (red[1] > 0.0 || !isfinite(fnorm)) { st = MOGE_ERR_NONFINITE; info = -1; }
        while (info == 0) {
            // ---- fdjac2 + qrfac (one column) -----------------------------------------------------
            double h = eps * fabs(x);
            if (h == 0.0) h = eps;
            double f_h = focal_fixed;
            if (!fixed) { focal_sums(P, x + h, red[0], red[1]); block_sum<2, REC_THREADS>(red, sh); f_h = red[0] / red[1]; }
            nfev++;
            const double xh = x + h;
            red[0] = 0.0;
            double c_rx[REC_PTS], c_ry[REC_PTS], c_jx[REC_PTS], c_jy[REC_PTS];     // residuals and Jacobian column of this thread's points: reused by the Householder pass below
#pragma unroll
            for (int i = 0; i < REC_PTS; i++)
                if (P.valid >> i & 1) {
                    const double d = (double)P.z[i] + x, dh = (double)P.z[i] + xh;
                    const double rx = f_cur * ((double)P.x[i] / d) - (double)P.u[i], ry = f_cur * ((double)P.y[i] / d) - (double)P.v[i];
                    const double hx = f_h * ((double)P.x[i] / dh) - (double)P.u[i], hy = f_h * ((double)P.y[i] / dh) - (double)P.v[i];
                    const double jx = (hx - rx) / h, jy = (hy - ry) / h;
                    c_rx[i] = rx; c_ry[i] = ry; c_jx[i] = jx; c_jy[i] = jy;
                    red[0] += jx * jx + jy * jy;
                    if (own_first && i == first_slot) { sh0[0] = jx; sh0[1] = rx; }
                }
            block_sum<1, REC_THREADS>(red, sh);          // (contains the barriers that publish sh0)
            const double j0 = sh0[0], fv0 = sh0[1];
            const double acnorm = sqrt(red[0]);
            double r = 0.0, qtf = fv0;
            if (acnorm != 0.0) {
                const double ajnorm = j0 < 0.0 ? -acnorm : acnorm;
                red[0] = 0.0;
#pragma unroll
                for (int i = 0; i < REC_PTS; i++)
                    if (P.valid >> i & 1) {
                        double vx = c_jx[i] / ajnorm;
                        if (own_first && i == first_slot) vx += 1.0;
                        red[0] += vx * c_rx[i] + (c_jy[i] / ajnorm) * c_ry[i];
                    }
                block_sum<1, REC_THREADS>(red, sh);
                r = -ajnorm;
                const double v0 = j0 / ajnorm + 1.0;
                qtf = v0 != 0.0 ? fv0 - red[0] : fv0;
            }
            if (iter == 1) { xnorm = fabs(diag * x); delta = factor * xnorm; if (delta == 0.0) delta = factor; }
            double gnorm = 0.0;
            if (fnorm != 0.0 && acnorm != 0.0) gnorm = fabs(r * (qtf / fnorm) / acnorm);
            if (gnorm <= gtol) { info = 4; break; }
            // ---- inner loop ------------------------------------------------------------------------
            for (;;) {
                double p;
                lmpar1(r, diag, qtf, delta, par, p);
                p = -p;
                const double x2 = x + p;
                const double pnorm = fabs(diag * p);
                if (iter == 1) delta = fmin(delta, pnorm);
                double f2 = focal_fixed;
                if (!fixed) { focal_sums(P, x2, red[0], red[1]); block_sum<2, REC_THREADS>(red, sh); f2 = red[0] / red[1]; }
                red[0] = 0.0;
#pragma unroll
                for (int i = 0; i < REC_PTS; i++)
                    if (P.valid >> i & 1) {
                        const double d = (double)P.z[i] + x2;
                        const double rx = f2 * ((double)P.x[i] / d) - (double)P.u[i], ry = f2 * ((double)P.y[i] / d) - (double)P.v[i];
                        red[0] += rx * rx + ry * ry;
                    }
                block_sum<1, REC_THREADS>(red, sh);
                nfev++;
                const double fnorm1 = sqrt(red[0]);
                double actred = -1.0;
                if (0.1 * fnorm1 < fnorm) { const double q = fnorm1 / fnorm; actred = 1.0 - q * q; }
                const double temp1 = fabs(r * p) / fnorm;
                const double temp2 = (sqrt(par) * pnorm) / fnorm;
                const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
                const double dirder = -(temp1 * temp1 + temp2 * temp2);
                const double ratio = prered != 0.0 ? actred / prered : 0.0;
                if (ratio <= 0.25) {
                    double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
                    if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                    delta = temp * fmin(delta, pnorm / 0.1);
                    par = par / temp;
                } else if (par == 0.0 || ratio >= 0.75) {
                    delta = pnorm / 0.5;
                    par = 0.5 * par;
                }
                if (ratio >= 1e-4) { x = x2; f_cur = f2; xnorm = fabs(diag * x); fnorm = fnorm1; iter++; }
                const bool c1 = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
                if (c1) info = 1;
                if (delta <= xtol * xnorm) info = 2;
                if (c1 && info == 2) info = 3;
                if (info != 0) break;
                if (nfev >= maxfev) info = 5;
                if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
                if (delta <= EPSMCH * xnorm) info = 7;
                if (gnorm <= EPSMCH) info = 8;
                if (info != 0) break;
                if (ratio >= 1e-4) break;
            }
        }
        if (st == 0) {
            shift_res = (float)x;
            if (!fixed) {
                // focal recomputed with the fp32 shift in fp32 (geometry_numpy.py:93-94); sums carried in fp64
                double num = 0.0, den = 0.0;
#pragma unroll
                for (int i = 0; i < REC_PTS; i++)
                    if (P.valid >> i & 1) {
                        const float d = P.z[i] + shift_res;
                        const float px = P.x[i] / d, py = P.y[i] / d;
                        num += (double)(px * P.u[i]) + (double)(py * P.v[i]);
                        den += (double)(px * px) + (double)(py * py);
                    }
                red[0] = num; red[1] = den;
                block_sum<2, REC_THREADS>(red, sh);
                focal_res = (float)((float)red[0] / (float)red[1]);
            }
        } else {
            shift_res = __builtin_nanf(""); focal_res = fixed ? focal_f : __builtin_nanf("");
        }
    }
    if (tid == 0) {
        if (focal_out) focal_out[b] = focal_res;
        if (shift_out) shift_out[b] = shift_res;
        if (intrinsics) {
            const float fx = focal_res / 2.f * fx_mul / fx_div, fy = focal_res / 2.f * fy_mul;
            float* K = intrinsics + b * 9;
            K[0] = fx; K[1] = 0.f; K[2] = 0.5f; K[3] = 0.f; K[4] = fy; K[5] = 0.5f; K[6] = 0.f; K[7] = 0.f; K[8] = 1.f;
        }
        if (st != 0 && status) atomicMin(status, st);
    }
}

int launch_recover(const float* points, const float* mask_prob, const uint8_t* mask_u8, const float* fov_deg, const float* focal_in, int B,
                   int H, int W, float* focal, float* shift, float* intrinsics, int* status, hipStream_t st, float mask_thr) {
    const double a = (double)W / (double)H;
    const double sx = a / sqrt(1 + a * a), sy = 1 / sqrt(1 + a * a);
    const float u0 = (float)(-sx * (W - 1) / W), u1 = (float)(sx * (W - 1) / W);
    const float v0 = (float)(-sy * (H - 1) / H), v1 = (float)(sy * (H - 1) / H);
    const float ustep = W > 1 ? (u1 - u0) / (float)(W - 1) : 0.f, vstep = H > 1 ? (v1 - v0) / (float)(H - 1) : 0.f;
    const float fov_c = (float)(a / sqrt(1 + a * a));
    const float diag = (float)sqrt(1 + a * a);
    if (moge_tune_get("REC_THREADS", 512) == 256)
        hipLaunchKernelGGL(recover_kernel<256>, dim3(B), dim3(256), 0, st, points, mask_prob, mask_u8, fov_deg, focal_in, H, W, u0, u1, ustep, v0, v1,
                           vstep, fov_c, diag, (float)a, diag, mask_thr, focal, shift, intrinsics, status);
    else if (moge_tune_get("REC_THREADS", 512) == 512)
        hipLaunchKernelGGL(recover_kernel<512>, dim3(B), dim3(512), 0, st, points, mask_prob, mask_u8, fov_deg, focal_in, H, W, u0, u1, ustep, v0, v1,
                           vstep, fov_c, diag, (float)a, diag, mask_thr, focal, shift, intrinsics, status);
    else
        hipLaunchKernelGGL(recover_kernel<1024>, dim3(B), dim3(1024), 0, st, points, mask_prob, mask_u8, fov_deg, focal_in, H, W, u0, u1, ustep, v0, v1,
                           vstep, fov_c, diag, (float)a, diag, mask_thr, focal, shift, intrinsics, status);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// v2.py:267-289 in one pass.  In-place safe (points_in may alias points_out, normal_in may alias normal_out).
__global__ __launch_bounds__(256) void finalize_kernel(const float* points_in, const float* normal_in,
                                                       const float* __restrict__ mask_prob, const float* __restrict__ metric,
                                                       const float* __restrict__ shift, const float* __restrict__ intr,
                                                       int B, int H, int W, int flags, float mask_thr, float* points_out, float* __restrict__ depth_out,
                                                       float* normal_out, uint8_t* __restrict__ mask_out) {
    // flags bit 8 (internal, MoGe-1): the validity mask is `mask > threshold` only - v1.py:358 has no `depth > 0` term (v2.py:268 has)
    const long total = (long)B * H * W;
    const float INF = __builtin_inff();
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ox = idx % W;
        const long t = idx / W;
        const int oy = t % H, b = t / H;
        // a model without a points head (v2.py:251-281, `points is None`): no depth, no intrinsics, no `depth > 0` term in the mask;
        // only the mask and the masked normal leave the kernel
        float x = 0.f, y = 0.f, z = 1.f;
        if (points_in) { x = points_in[idx * 3]; y = points_in[idx * 3 + 1]; z = points_in[idx * 3 + 2] + shift[b]; }
        bool m = true;
        if (mask_prob) m = (mask_prob[idx] > mask_thr) && ((flags & 0x100) || z > 0.f);
        float depth = z;
        if (points_in && (flags & MOGE_FORCE_PROJECTION)) {
            const float fx = intr[b * 9], fy = intr[b * 9 + 4], cx = intr[b * 9 + 2], cy = intr[b * 9 + 5];
            const float u = ((float)ox + 0.5f) / (float)W, v = ((float)oy + 0.5f) / (float)H;
            x = (u - cx) / fx * depth;
            y = (v - cy) / fy * depth;
        }
        if (metric && points_in) { const float s = metric[b]; x *= s; y *= s; z *= s; depth *= s; }
        float nx = 0.f, ny = 0.f, nz = 0.f;
        if (normal_in) { nx = normal_in[idx * 3]; ny = normal_in[idx * 3 + 1]; nz = normal_in[idx * 3 + 2]; }
        if ((flags & MOGE_APPLY_MASK) && mask_prob && !m) { x = INF; y = INF; z = INF; depth = INF; nx = 0.f; ny = 0.f; nz = 0.f; }
        if (points_out && points_in) { points_out[idx * 3] = x; points_out[idx * 3 + 1] = y; points_out[idx * 3 + 2] = z; }
        if (depth_out && points_in) depth_out[idx] = depth;
        if (normal_out && normal_in) { normal_out[idx * 3] = nx; normal_out[idx * 3 + 1] = ny; normal_out[idx * 3 + 2] = nz; }
        if (mask_out && mask_prob) mask_out[idx] = m ? 1 : 0;
    }
}
int launch_finalize(const float* points_in, const float* normal_in, const float* mask_prob, const float* metric, const float* shift,
                    const float* intr, int B, int H, int W, int flags, float* points_out, float* depth_out, float* normal_out,
                    uint8_t* mask_out, hipStream_t st, float mask_thr) {
    const long total = (long)B * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(256), 0, st, points_in, normal_in, mask_prob, metric, shift, intr, B, H, W, flags, mask_thr,
                       points_out, depth_out, normal_out, mask_out);
    return (int)hipGetLastError();
}

// --------------------------------------------------------------------------------------------
// Caller-side mesh clean-up (scripts/infer.py:127: `mask & ~utils3d.np.depth_map_edge(depth, rtol=threshold)`).  utils3d is an un-vendored
// dependency (pyproject.toml:23); its published algorithm, restated: diff = maxpool3x3(depth) + maxpool3x3(-depth) with -inf padding
// (= max - min over the in-image 3x3 neighbourhood), edge = diff / depth > rtol evaluated in IEEE arithmetic (inf / inf = nan -> false,
// inf / finite = inf -> true, so a valid pixel next to a masked-out (+inf) one is an edge).  out = mask & ~edge, one byte per pixel.
// --------------------------------------------------------------------------------------------
__global__ void depth_edge_mask_kernel(const float* __restrict__ depth, const unsigned char* __restrict__ mask, unsigned char* __restrict__ out,
                                       long B, int H, int W, float rtol) {
    const long px = (long)H * W, total = B * px;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / px, p = idx - b * px;
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
        const float* d = depth + b * px;
        float mx = -INFINITY, mn = INFINITY;
        for (int dy = -1; dy <= 1; dy++) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -1; dx <= 1; dx++) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const float v = d[(long)yy * W + xx];
                mx = fmaxf(mx, v);            // fmaxf drops a NaN operand like np.fmax; depth maps on this path carry no NaN (finite or +inf)
                mn = fminf(mn, v);
            }
        }
        const float diff = mx + (-mn);
        const bool edge = (diff / d[p]) > rtol;
        out[idx] = (mask ? mask[idx] != 0 : true) && !edge;
    }
}
int launch_depth_edge_mask(const float* depth, const unsigned char* mask, unsigned char* out, int B, int H, int W, float rtol, hipStream_t st) {
    const long total = (long)B * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(depth_edge_mask_kernel, dim3(blocks), dim3(256), 0, st, depth, mask, out, (long)B, H, W, rtol);
    return (int)hipGetLastError();
}
