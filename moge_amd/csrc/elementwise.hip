// Bandwidth-bound kernels of the encoder side: image resize+normalise+patchify, position-embedding bicubic
// resample, cls-row init, LayerNorm.  All fp32 math; storage type T on the GEMM-facing side.
#include "common.h"

// --------------------------------------------------------------------------------------------
// K0: antialiased bilinear resize (F.interpolate(..., mode="bilinear", antialias=True), modules.py:121) of the
// input image to (14*rows, 14*cols), ImageNet normalisation (modules.py:122), written directly in the im2col layout
// the patch-embed GEMM wants: A0[(b*Np + py*cols + px)][c*196 + iy*14 + ix], row stride ldk (zero padded to ldk).
// ATen's _upsample_bilinear2d_aa: triangle filter of support max(scale,1), weights normalised per output pixel.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void aa_range(int o, float scale, int in_size, int& lo, int& n, float& center, float& invscale, float& support) {
    support = scale >= 1.f ? scale : 1.f;
    invscale = scale >= 1.f ? 1.f / scale : 1.f;
    center = scale * (o + 0.5f);
    lo = (int)(center - support + 0.5f);
    lo = lo < 0 ? 0 : lo;
    int hi = (int)(center + support + 0.5f);
    hi = hi > in_size ? in_size : hi;
    n = hi - lo;
}
__device__ __forceinline__ float tri(float x) { x = fabsf(x); return x < 1.f ? 1.f - x : 0.f; }

constexpr int PATCH_K = 3 * 14 * 14;       // im2col row of a 14 x 14 RGB patch (model.hip KPATCH); the row pitch ldk pads it to a multiple of 64
template <typename TIn, typename TOut>
__global__ void preprocess_kernel(const TIn* __restrict__ img, TOut* __restrict__ out, int B, int H, int W, int rows, int cols,
                                  int ldk, int nchw_out, int round16, int aa, float m0, float m1, float m2, float s0, float s1, float s2,
                                  int* __restrict__ zero_i32, int zero_n) {
    // aa == 0: onnx_compatible_mode (modules.py:121 antialias=False): ATen upsample_bilinear2d, align_corners=False - two taps per axis at
    // src = scale * (dst + 0.5) - 0.5 clamped at 0, i1 = min(i0 + 1, in - 1), whatever the scale
    // round16: fp32 input whose values are first rounded to fp16 - the reference's `image.to(dtype=self.dtype)` for a .half() model
    // (v2.py:229) - done here on load instead of as a separate cast pass over the image
    // one thread per output PIXEL: the filter ranges and weight sums depend on (oy, ox) only and serve the three channels
    // (per-channel arithmetic and its order are unchanged: horizontal pass first, then vertical, fp32)
    // Round 6 (one image: every launch in front of the first block is ~4-5 us of dispatch): the im2col matrix's K padding columns [588, ldk) are zeroed HERE - the
    // pixel (iy, ix) of a patch zeroes padding column iy * 14 + ix of its row (ldk - 588 <= 196) - instead of by zero_cols_kernel, and so are the forward's
    // device-side counters (zero_i32[0 .. zero_n): the fused LN finalize's row-block counters) instead of by a hipMemsetAsync (two fill kernels).
    const int OH = rows * 14, OW = cols * 14;
    const long total = (long)B * OH * OW;
    const float scale_y = (float)H / (float)OH, scale_x = (float)W / (float)OW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < zero_n; i += (long)gridDim.x * blockDim.x) zero_i32[i] = 0;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = idx % OW;
        long t = idx / OW;
        const int oy = t % OH;
        const int b = t / OH;
        if (!nchw_out) {
            const int py = oy / 14, px = ox / 14;
            const int j = PATCH_K + (oy - py * 14) * 14 + (ox - px * 14);
            if (j < ldk) out[((size_t)b * rows * cols + (size_t)py * cols + px) * ldk + j] = (TOut)0.f;
        }
        if (!aa) {
            float sy = scale_y * ((float)oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
            float sx = scale_x * ((float)ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
            const int py = oy / 14, iy = oy - py * 14, px = ox / 14, ix = ox - px * 14;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const TIn* pl = img + ((size_t)b * 3 + c) * H * W;
                float v00 = (float)pl[(size_t)y0 * W + x0], v01 = (float)pl[(size_t)y0 * W + x1], v10 = (float)pl[(size_t)y1 * W + x0], v11 = (float)pl[(size_t)y1 * W + x1];
                if (round16) { v00 = (float)(f16)v00; v01 = (float)(f16)v01; v10 = (float)(f16)v10; v11 = (float)(f16)v11; }
                const float r = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
                const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
                const float v = (r - mean) / sd;
                if (nchw_out) out[(((size_t)b * 3 + c) * OH + oy) * OW + ox] = (TOut)v;
                else out[((size_t)b * rows * cols + (size_t)py * cols + px) * ldk + c * 196 + iy * 14 + ix] = (TOut)v;
            }
            continue;
        }
        int ylo, yn, xlo, xn; float yc, yis, ysup, xc, xis, xsup;
        aa_range(oy, scale_y, H, ylo, yn, yc, yis, ysup);
        aa_range(ox, scale_x, W, xlo, xn, xc, xis, xsup);
        float wysum = 0.f, wxsum = 0.f;
        for (int j = 0; j < yn; j++) wysum += tri((j + ylo - yc + 0.5f) * yis);
        for (int i = 0; i < xn; i++) wxsum += tri((i + xlo - xc + 0.5f) * xis);
        float acc[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < yn; j++) {
            const float wy = tri((j + ylo - yc + 0.5f) * yis) / wysum;
            float h[3] = {0.f, 0.f, 0.f};
            for (int i = 0; i < xn; i++) {
                const float wx = tri((i + xlo - xc + 0.5f) * xis) / wxsum;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float pv = (float)img[(((size_t)b * 3 + c) * H + (ylo + j)) * W + xlo + i];
                    if (round16) pv = (float)(f16)pv;
                    h[c] += wx * pv;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) acc[c] += wy * h[c];
        }
        const int py = oy / 14, iy = oy - py * 14, px = ox / 14, ix = ox - px * 14;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
            const float v = (acc[c] - mean) / sd;
            if (nchw_out) out[(((size_t)b * 3 + c) * OH + oy) * OW + ox] = (TOut)v;
            else out[((size_t)b * rows * cols + (size_t)py * cols + px) * ldk + c * 196 + iy * 14 + ix] = (TOut)v;
        }
    }
}

template <typename TIn, typename TOut>
int launch_preprocess(const void* img, void* out, int B, int H, int W, int rows, int cols, int ldk, int nchw_out, int round16, int aa,
                      const float* mean, const float* std_, hipStream_t st, int* zero_i32, int zero_n) {
    if (!nchw_out && (ldk < PATCH_K || ldk - PATCH_K > 196)) return -1;         // (the padding columns are zeroed by the patch's own 196 pixels)
    const long total = (long)B * rows * 14 * cols * 14;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL((preprocess_kernel<TIn, TOut>), dim3(blocks), dim3(256), 0, st, (const TIn*)img, (TOut*)out, B, H, W, rows, cols,
                       ldk, nchw_out, round16, aa, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], zero_i32, zero_i32 ? zero_n : 0);
    return (int)hipGetLastError();
}
template int launch_preprocess<float, f16>(const void*, void*, int, int, int, int, int, int, int, int, int, const float*, const float*, hipStream_t, int*, int);
template int launch_preprocess<float, float>(const void*, void*, int, int, int, int, int, int, int, int, int, const float*, const float*, hipStream_t, int*, int);
template int launch_preprocess<f16, f16>(const void*, void*, int, int, int, int, int, int, int, int, int, const float*, const float*, hipStream_t, int*, int);
template int launch_preprocess<f16, float>(const void*, void*, int, int, int, int, int, int, int, int, int, const float*, const float*, hipStream_t, int*, int);

// Caller-side ingest (scripts/infer.py:98: `torch.tensor(image / 255, dtype=torch.float32).permute(2, 0, 1)`, then v2.py:229 casts to the model
// dtype): uint8 (B,H,W,3) -> T (B,3,H,W).  numpy divides in float64 and the tensor constructor rounds to float32 once: same here.
template <typename T>
__global__ void u8hwc_to_chw_kernel(const unsigned char* __restrict__ in, T* __restrict__ out, long B, int H, int W) {
    const long px = (long)H * W, total = B * px;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / px, p = idx - b * px;
        const unsigned char* s = in + idx * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) out[(b * 3 + c) * px + p] = (T)(float)((double)s[c] / 255.0);
    }
}
template <typename T>
int launch_u8hwc_to_chw(const void* in, void* out, int B, int H, int W, hipStream_t st) {
    const long total = (long)B * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(u8hwc_to_chw_kernel<T>, dim3(blocks), dim3(256), 0, st, (const unsigned char*)in, (T*)out, (long)B, H, W);
    return (int)hipGetLastError();
}
template int launch_u8hwc_to_chw<f16>(const void*, void*, int, int, int, hipStream_t);
template int launch_u8hwc_to_chw<float>(const void*, void*, int, int, int, hipStream_t);

// (The K padding columns [588, ldk) of the im2col matrix are zeroed by preprocess_kernel itself since round 6.)

// --------------------------------------------------------------------------------------------
// Position embedding for an (rows x cols) grid: bicubic (A=-0.75, align_corners=False, no antialias) resample of the
// 37x37 pre-training grid with the reference's scale_factor=(n+0.1)/37 kludge (vision_transformer.py:187-221):
// src = (dst+0.5)*rscale - 0.5 with rscale = float(1/scale_factor); taps clamp-indexed.  Row 0 = cls position.
// Bypassed (plain copy) iff rows == cols == 37.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
    const float A = -0.75f;
    float x = t + 1.f;
    w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
    x = t;
    w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 1.f - t;
    w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 2.f - t;
    w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

__global__ void posembed_kernel(const float* __restrict__ pos, float* __restrict__ out, int D, int M, int rows, int cols,
                                float rscale_y, float rscale_x, int bypass) {
    const long total = (long)(1 + rows * cols) * D;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int d = idx % D;
        const int p = idx / D;
        if (p == 0 || bypass) { out[idx] = pos[idx]; continue; }
        const int oy = (p - 1) / cols, ox = (p - 1) - oy * cols;
        const float sy = rscale_y * (oy + 0.5f) - 0.5f, sx = rscale_x * (ox + 0.5f) - 0.5f;
        const float fy = floorf(sy), fx = floorf(sx);
        const int iy = (int)fy, ix = (int)fx;
        float wy[4], wx[4];
        cubic_w(sy - fy, wy);
        cubic_w(sx - fx, wx);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int yy = iy - 1 + j; yy = yy < 0 ? 0 : (yy > M - 1 ? M - 1 : yy);
            float r = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int xx = ix - 1 + i; xx = xx < 0 ? 0 : (xx > M - 1 ? M - 1 : xx);
                r += wx[i] * pos[(size_t)(1 + yy * M + xx) * D + d];
            }
            acc += wy[j] * r;
        }
        out[idx] = acc;
    }
}
// size_mode (onnx_compatible_mode, vision_transformer.py:192,202-210): never bypassed, resampled by OUTPUT SIZE: src scale = 37 / n in float
// (ATen area_pixel_compute_scale without a scale factor) instead of 1 / ((n + 0.1) / 37)
int launch_posembed(const float* pos, float* out, int D, int rows, int cols, int size_mode, hipStream_t st) {
    const int M = 37;
    const int bypass = (!size_mode && rows == M && cols == M) ? 1 : 0;
    float ry = (float)(1.0 / ((double)(rows + 0.1) / M)), rx = (float)(1.0 / ((double)(cols + 0.1) / M));
    if (size_mode) { ry = (float)M / (float)rows; rx = (float)M / (float)cols; }
    const long total = (long)(1 + rows * cols) * D;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(posembed_kernel, dim3(blocks), dim3(256), 0, st, pos, out, D, M, rows, cols, ry, rx, bypass);
    return (int)hipGetLastError();
}

// x[b, 0, :] = cls_token + pos[0]   (vision_transformer.py:230-231)
// (cls row = cls_token + pos_embed[0]: written by the patch-embed GEMM's EPI_PATCH epilogue since round 6, gemm.hip.)

// --------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-6, vision_transformer.py:95): one wave per row of the fp32 residual stream, two-pass in
// registers.  D % 128 == 0, D <= 1024 (ViT-S/B/L).  Output modes:
//   plain: out[row*ldo + col]
//   tap  : (get_intermediate_layers, vision_transformer.py:321-324) token 0 -> cls_out[b*D + col] (fp32, optional),
//          token t>0 -> out[(b*Np + t-1)*ldo + coloff + col]  (the K-concatenated output-projection operand)
// --------------------------------------------------------------------------------------------
// LN_RPW: consecutive rows per wave - 4 for large row counts (weights / bias loaded once per 4 rows), 1 when there are few rows (one image: 3601 rows = 226 workgroups of
// 16 rows left most of the chip idle and every wave walked four dependent load -> reduce -> store round trips; 901 workgroups of 4 rows: 13.2 -> see EXPERIMENTS R6.10)
constexpr long LN_FEW_ROWS = 16384;      // up to ~4 images of 3601 tokens: one row per wave
template <typename T, typename TX = float, int LN_RPW = 4>      // TX: element type of the residual stream (float; f16 for `.half()` models, whose stream is fp16)
__global__ __launch_bounds__(256) void layernorm_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ out, float* __restrict__ cls_out, long rowsN, int D, int ldo,
                                                        int coloff, int tap_mode, int Ntok) {
    // One wave per row, LN_RPW consecutive rows per wave: a lane owns columns i*256 + 4*lane .. +3 (16-byte loads: a wave
    // instruction moves 1 KiB of the fp32 row, 8-byte stores of the storage type), weight / bias stay in registers.
    const int lane = threadIdx.x & 63;
    const long row0 = (blockIdx.x * 4L + (threadIdx.x >> 6)) * LN_RPW;
    if (row0 >= rowsN) return;
    const int nit = (D + 255) >> 8;
    f32x4 ww[4], bb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int col = i * 256 + lane * 4;
        if (i < nit && col < D) { ww[i] = *reinterpret_cast<const f32x4*>(w + col); bb[i] = *reinterpret_cast<const f32x4*>(bias + col); }
    }
    const float invD = 1.f / (float)D;
    for (int r = 0; r < LN_RPW; r++) {
        const long row = row0 + r;
        if (row >= rowsN) break;
        const TX* xr = x + row * (long)D;
        f32x4 v[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int col = i * 256 + lane * 4;
            if (i < nit && col < D) {
                float t[4];
                load4(xr + col, t);
                v[i] = f32x4{t[0], t[1], t[2], t[3]};
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int col = i * 256 + lane * 4;
            if (i < nit && col < D) {
#pragma unroll
                for (int e = 0; e < 4; e++) { const float a = v[i][e] - mean; q += a * a; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q * invD + 1e-6f);
        T* op = nullptr; float* cp = nullptr;
        if (tap_mode) {
            const long b = row / Ntok; const int t = (int)(row - b * Ntok);
            if (t == 0) { if (!cls_out) continue; cp = cls_out + b * D; }
            else op = out + (b * (Ntok - 1) + t - 1) * (long)ldo + coloff;
        } else {
            op = out + row * (long)ldo + coloff;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int col = i * 256 + lane * 4;
            if (i < nit && col < D) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; e++) y[e] = fmaf((v[i][e] - mean) * rstd, ww[i][e], bb[i][e]);
                if (cp) *reinterpret_cast<f32x4*>(cp + col) = f32x4{y[0], y[1], y[2], y[3]};
                else store4(op + col, y[0], y[1], y[2], y[3]);
            }
        }
    }
}
// --------------------------------------------------------------------------------------------
// LayerNorm folded into the neighbouring GEMMs (fp16 path, model.hip "LN fold").
//   ln_raw_kernel      : first block only - fp16 copy of the residual row + its (mean, rstd) (two-pass, as layernorm_kernel)
//   ln_finalize_kernel : (sum, sum of squares) partials of the RESID epilogues -> (mean, rstd); var = E[x^2] - mean^2 in fp32, clamped at 0
//   fold_ln_kernel     : W'[n][k] = T(g[k] * W[n][k]),  c[n] = sum_k float(W'[n][k]),  b'[n] = b[n] + sum_k beta[k] * W[n][k]
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ln_raw_kernel(const float* __restrict__ x, T* __restrict__ out, float* __restrict__ mr, long rowsN, int D) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rowsN) return;
    const int nit = (D + 255) >> 8;
    const float invD = 1.f / (float)D;
    const float* xr = x + row * (long)D;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int col = i * 256 + lane * 4;
        if (i < nit && col < D) { v[i] = *reinterpret_cast<const f32x4*>(xr + col); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int col = i * 256 + lane * 4;
        if (i < nit && col < D) {
#pragma unroll
            for (int e = 0; e < 4; e++) { const float a = v[i][e] - mean; q += a * a; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * invD + 1e-6f);
    if (lane == 0) *reinterpret_cast<f32x2*>(mr + 2 * row) = f32x2{mean, rstd};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int col = i * 256 + lane * 4;
        if (i < nit && col < D) store4(out + row * (long)D + col, v[i][0], v[i][1], v[i][2], v[i][3]);
    }
}
template <typename T>
int launch_ln_raw(const float* x, void* out, float* mr, long rowsN, int D, hipStream_t st) {
    if (D % 4 != 0 || D > 1024) return -1;
    hipLaunchKernelGGL(ln_raw_kernel<T>, dim3((unsigned)((rowsN + 3) / 4)), dim3(256), 0, st, x, (T*)out, mr, rowsN, D);
    return (int)hipGetLastError();
}
template int launch_ln_raw<f16>(const float*, void*, float*, long, int, hipStream_t);

__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, float* __restrict__ mr, long rowsN, int NP, float invD) {
    // 8 lanes per row: lane j sums the 16-byte quads (two (s1, s2) pairs) j, j + 8, ... of the row's NP pairs - a row's 8 NP bytes are read as
    // full 128-byte segments (one thread per row read 64 different lines per instruction: 13 us per launch at batch 32) - then a 1-2-4
    // butterfly.  One fixed summation order for every batch size (this kernel is the only finaliser), so batch items stay independent of their batch.
    const long gid = blockIdx.x * 256L + threadIdx.x;
    const long row = gid >> 3;
    const int j = (int)(gid & 7);
    float s1 = 0.f, s2 = 0.f;
    if (row < rowsN) {
        const f32x4* p = reinterpret_cast<const f32x4*>(part + row * (long)NP * 2);
        for (int q = j; q < NP / 2; q += 8) {
            const f32x4 t = p[q];
            s1 += t[0]; s2 += t[1];
            s1 += t[2]; s2 += t[3];
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (row < rowsN && j == 0) {
        const float mean = s1 * invD;
        const float var = fmaxf(fmaf(-mean, mean, s2 * invD), 0.f);
        *reinterpret_cast<f32x2*>(mr + 2 * row) = f32x2{mean, rsqrtf(var + 1e-6f)};
    }
}
int launch_ln_finalize(const float* part, float* mr, long rowsN, int NP, int D, hipStream_t st) {
    if (NP & 1) return -1;
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((rowsN * 8 + 255) / 256)), dim3(256), 0, st, part, mr, rowsN, NP, 1.f / (float)D);
    return (int)hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void fold_ln_kernel(const float* __restrict__ W, const float* __restrict__ g, const float* __restrict__ beta,
                                                      const float* __restrict__ b, T* __restrict__ Wf, float* __restrict__ c, float* __restrict__ bf, int K) {
    __shared__ float red[2][256];
    const int n = blockIdx.x, t = threadIdx.x;
    float sc = 0.f, sb = 0.f;
    for (int k = t; k < K; k += 256) {
        const float w = W[(size_t)n * K + k];
        const T wf = (T)(g[k] * w);
        Wf[(size_t)n * K + k] = wf;
        sc += (float)wf;
        sb = fmaf(beta[k], w, sb);
    }
    red[0][t] = sc; red[1][t] = sb;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; }
        __syncthreads();
    }
    if (t == 0) { c[n] = red[0][0]; bf[n] = b[n] + red[1][0]; }
}
template <typename T>
int launch_fold_ln(const float* W, const float* g, const float* beta, const float* b, void* Wf, float* c, float* bf, int N, int K, hipStream_t st) {
    hipLaunchKernelGGL(fold_ln_kernel<T>, dim3((unsigned)N), dim3(256), 0, st, W, g, beta, b, (T*)Wf, c, bf, K);
    return (int)hipGetLastError();
}
template int launch_fold_ln<f16>(const float*, const float*, const float*, const float*, void*, float*, float*, int, int, hipStream_t);

#ifdef MOGE_EXPERIMENTS
#include "../../tools/experiments/layernorm_lr_exp.inc"     // low-register LayerNorm (a measured, rejected co-residency experiment)
#endif

template <typename T>
int launch_layernorm(const float* x, const float* w, const float* b, void* out, float* cls_out, long rowsN, int D, int ldo, int coloff,
                     int tap_mode, int Ntok, hipStream_t st) {
    if (D % 4 != 0 || D > 1024 || (ldo & 3) || (coloff & 3)) return -1;
#ifdef MOGE_EXPERIMENTS
    if (sizeof(T) == 2 && moge_tune_get("LN_LOWREG", 0)) {
        hipLaunchKernelGGL(layernorm_lr_kernel<T>, dim3((unsigned)((rowsN + 15) / 16)), dim3(256), 0, st, x, w, b, (T*)out, cls_out, rowsN, D, ldo, coloff,
                           tap_mode, Ntok);
        return (int)hipGetLastError();
    }
#endif
    if (rowsN <= LN_FEW_ROWS) {
        hipLaunchKernelGGL((layernorm_kernel<T, float, 1>), dim3((unsigned)((rowsN + 3) / 4)), dim3(256), 0, st, x, w, b, (T*)out, cls_out, rowsN, D, ldo, coloff, tap_mode, Ntok);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(layernorm_kernel<T>, dim3((unsigned)((rowsN + 15) / 16)), dim3(256), 0, st, x, w, b, (T*)out, cls_out, rowsN, D, ldo, coloff,
                       tap_mode, Ntok);
    return (int)hipGetLastError();
}
template int launch_layernorm<f16>(const float*, const float*, const float*, void*, float*, long, int, int, int, int, int, hipStream_t);
template int launch_layernorm<float>(const float*, const float*, const float*, void*, float*, long, int, int, int, int, int, hipStream_t);

// the same LayerNorm on an fp16 residual stream (`.half()` models: the final-norm taps, vision_transformer.py:322)
int launch_layernorm_x16(const void* x16, const float* w, const float* b, void* out, float* cls_out, long rowsN, int D, int ldo, int coloff,
                         int tap_mode, int Ntok, hipStream_t st) {
    if (D % 4 != 0 || D > 1024 || (ldo & 3) || (coloff & 3)) return -1;
    if (rowsN <= LN_FEW_ROWS) {
        hipLaunchKernelGGL((layernorm_kernel<f16, f16, 1>), dim3((unsigned)((rowsN + 3) / 4)), dim3(256), 0, st, (const f16*)x16, w, b, (f16*)out, cls_out, rowsN, D, ldo, coloff, tap_mode, Ntok);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL((layernorm_kernel<f16, f16>), dim3((unsigned)((rowsN + 15) / 16)), dim3(256), 0, st, (const f16*)x16, w, b, (f16*)out, cls_out, rowsN, D, ldo,
                       coloff, tap_mode, Ntok);
    return (int)hipGetLastError();
}

// --------------------------------------------------------------------------------------------
// small helpers: fp32 <-> storage conversions (weight packing, test entry points, debug taps)
// --------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void convert_kernel(const TS* s, TD* d, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = (TD)(float)s[i];
}
template <typename TS, typename TD>
int launch_convert(const void* s, void* d, long n, hipStream_t st) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL((convert_kernel<TS, TD>), dim3(blocks), dim3(256), 0, st, (const TS*)s, (TD*)d, n);
    return (int)hipGetLastError();
}
template int launch_convert<float, f16>(const void*, void*, long, hipStream_t);
template int launch_convert<float, float>(const void*, void*, long, hipStream_t);
template int launch_convert<f16, float>(const void*, void*, long, hipStream_t);

// generic strided repack: dst[o0*ds0 + o1*ds1 + o2*ds2 + o3] = src[o0*ss0 + o1*ss1 + o2*ss2 + o3*ss3], dims (n0,n1,n2,n3)
template <typename TD>
__global__ void repack_kernel(const float* src, TD* dst, int n0, int n1, int n2, int n3, long ss0, long ss1, long ss2, long ss3,
                              long ds0, long ds1, long ds2) {
    const long total = (long)n0 * n1 * n2 * n3;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int i3 = t % n3; t /= n3;
        const int i2 = t % n2; t /= n2;
        const int i1 = t % n1; t /= n1;
        const int i0 = (int)t;
        dst[i0 * ds0 + i1 * ds1 + i2 * ds2 + i3] = (TD)src[i0 * ss0 + i1 * ss1 + i2 * ss2 + i3 * ss3];
    }
}

// --------------------------------------------------------------------------------------------
// CT3 (conv_pp.hip, round 6): composed weights of ConvTranspose2d(k2, s2) + 3x3.  P[(k * Cout + co)][s * Cin + ci] (fp32) holds the 36 basic products
// W3[:, :, k] . WT[:, :, s]^T (k = ky * 3 + kx, s = sy * 2 + sx; one fp32 GEMM in model.hip); every composed matrix - the 16 (phase, tap) blocks of the interior
// conv, the 12 x 2 (border class, cell) blocks of the border correction - is a short signed sum of them, listed per output block in a term table built on
// the host (model.hip ct3_slots).  One rounding to fp16 per composed weight.
// --------------------------------------------------------------------------------------------
__global__ void ct3_combine_kernel(const float* __restrict__ P, const Ct3Slot* __restrict__ slots, int Cout, int Cin, f16* __restrict__ out, int ld) {
    const Ct3Slot sl = slots[blockIdx.y];
    const long total = (long)Cout * Cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i / Cin), ci = (int)(i - (long)co * Cin);
        float a = 0.f;
        for (int t = 0; t < sl.nterms; t++) {
            const int term = sl.term[t], k = term & 15, sidx = (term >> 4) & 3;
            const float v = P[((size_t)k * Cout + co) * (4 * (size_t)Cin) + (size_t)sidx * Cin + ci];
            a += (term & 256) ? -v : v;
        }
        out[((size_t)sl.rowblk * Cout + co) * ld + (size_t)sl.colblk * Cin + ci] = (f16)a;
    }
}
int launch_ct3_combine(const float* P, const Ct3Slot* slots_dev, int nslots, int Cout, int Cin, void* out, int ld, hipStream_t st) {
    if (nslots <= 0) return 0;
    hipLaunchKernelGGL(ct3_combine_kernel, dim3((unsigned)(((long)Cout * Cin + 255) / 256), (unsigned)nslots), dim3(256), 0, st, P, slots_dev, Cout, Cin, (f16*)out, ld);
    return (int)hipGetLastError();
}

template <typename TD>
int launch_repack(const float* src, void* dst, int n0, int n1, int n2, int n3, long ss0, long ss1, long ss2, long ss3, long ds0, long ds1,
                  long ds2, hipStream_t st) {
    const long total = (long)n0 * n1 * n2 * n3;
    if (total <= 0) return 0;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(repack_kernel<TD>, dim3(blocks), dim3(256), 0, st, src, (TD*)dst, n0, n1, n2, n3, ss0, ss1, ss2, ss3, ds0, ds1, ds2);
    return (int)hipGetLastError();
}
template int launch_repack<f16>(const float*, void*, int, int, int, int, long, long, long, long, long, long, long, hipStream_t);
template int launch_repack<float>(const float*, void*, int, int, int, int, long, long, long, long, long, long, long, hipStream_t);

// --------------------------------------------------------------------------------------------
// bilinear x2 upsample (align_corners=False) followed by a replicate-padded 3x3 conv (Resampler 'bilinear',
// modules.py:155-159) == four 3x3 convs on the LOW-res map, one per output parity (py,px), with replicate clamping of the
// low-res indices (the clamped bilinear taps and the replicate pad of the virtual hi-res map coincide with index
// clamping).  hi-res row 2i   = .25 x[i-1] + .75 x[i],   hi-res row 2i+1 = .75 x[i] + .25 x[i+1], so with
//   R[0] = [[.75,.25,0],[.25,.75,0],[0,.75,.25]]   R[1] = [[.25,.75,0],[0,.75,.25],[0,.25,.75]]     (R[p][dy+1][a+1])
//   Weff[(py,px,co)][(a,b),ci] = sum_{dy,dx} W[co][ci][dy][dx] * R[py][dy][a] * R[px][dx][b].
// dst: [(py*2+px)*Cout + co][(a*3+b)*Cin + ci], torch src: [co][ci][3][3].  Combination in fp32, one rounding to T.
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_phase_conv_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int nearest) {
    // nearest (Resampler 'nearest', modules.py:152-156): hi-res rows 2i and 2i + 1 are both x[i], so tap dy of output parity p reads low-res row
    // i + floor((p + dy - 1) / 2): R[0] = [[1,0,0],[0,1,0],[0,1,0]], R[1] = [[0,1,0],[0,1,0],[0,0,1]]; the replicate pad again coincides with index clamping
    const float RB[2][3][3] = {{{.75f, .25f, 0.f}, {.25f, .75f, 0.f}, {0.f, .75f, .25f}}, {{.25f, .75f, 0.f}, {0.f, .75f, .25f}, {0.f, .25f, .75f}}};
    const float RN[2][3][3] = {{{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 1.f, 0.f}}, {{0.f, 1.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}}};
    const float (*R)[3][3] = nearest ? RN : RB;
    const long total = 4L * Cout * 9 * Cin;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long t = idx;
        const int ci = t % Cin; t /= Cin;
        const int tap = t % 9; t /= 9;
        const int co = t % Cout; t /= Cout;
        const int ph = (int)t;
        const int py = ph >> 1, px = ph & 1, a = tap / 3, b = tap % 3;
        const float* ws = w + ((size_t)co * Cin + ci) * 9;
        float acc = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) acc += ws[dy * 3 + dx] * (R[py][dy][a] * R[px][dx][b]);
        dst[idx] = (T)acc;
    }
}
template <typename T>
int launch_pack_phase_conv(const float* w, void* dst, int Cout, int Cin, hipStream_t st, int nearest) {
    const long total = 4L * Cout * 9 * Cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_phase_conv_kernel<T>, dim3(blocks), dim3(256), 0, st, w, (T*)dst, Cout, Cin, nearest);
    return (int)hipGetLastError();
}
template int launch_pack_phase_conv<f16>(const float*, void*, int, int, hipStream_t, int);
template int launch_pack_phase_conv<float>(const float*, void*, int, int, hipStream_t, int);

// ============================================================================================================================
// MoGe-1 (moge/model/v1.py) support kernels
// ============================================================================================================================
// v1.py:275: F.interpolate(image, (rh, rw), mode="bicubic", align_corners=False, antialias=True) - ATen _upsample_bicubic2d_aa: cubic
// convolution filter with a = -0.5, support 2 * max(scale, 1), weights normalised per output pixel; horizontal pass, then vertical (same
// index / weight formulas as aa_range above with the wider support).  NCHW in (TIn) -> NCHW fp32 out.
__device__ __forceinline__ float cubic_aa(float x) {
    constexpr float a = -0.5f;
    x = fabsf(x);
    if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
    if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
    return 0.f;
}
__device__ __forceinline__ void aa_range_cubic(int o, float scale, int in_size, int& lo, int& n, float& center, float& invscale) {
    const float support = scale >= 1.f ? 2.f * scale : 2.f;
    invscale = scale >= 1.f ? 1.f / scale : 1.f;
    center = scale * (o + 0.5f);
    lo = (int)(center - support + 0.5f);
    lo = lo < 0 ? 0 : lo;
    int hi = (int)(center + support + 0.5f);
    hi = hi > in_size ? in_size : hi;
    n = hi - lo;
}
template <typename TIn>
__global__ void resize_bicubic_aa_kernel(const TIn* __restrict__ img, float* __restrict__ out, int B, int H, int W, int OH, int OW, int round16) {
    const long total = (long)B * OH * OW;
    const float scale_y = (float)H / (float)OH, scale_x = (float)W / (float)OW;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = idx % OW;
        long t = idx / OW;
        const int oy = t % OH;
        const int b = t / OH;
        int ylo, yn, xlo, xn; float yc, yis, xc, xis;
        aa_range_cubic(oy, scale_y, H, ylo, yn, yc, yis);
        aa_range_cubic(ox, scale_x, W, xlo, xn, xc, xis);
        float wysum = 0.f, wxsum = 0.f;
        for (int j = 0; j < yn; j++) wysum += cubic_aa((j + ylo - yc + 0.5f) * yis);
        for (int i = 0; i < xn; i++) wxsum += cubic_aa((i + xlo - xc + 0.5f) * xis);
        float acc[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < yn; j++) {
            const float wy = cubic_aa((j + ylo - yc + 0.5f) * yis) / wysum;
            float h[3] = {0.f, 0.f, 0.f};
            for (int i = 0; i < xn; i++) {
                const float wx = cubic_aa((i + xlo - xc + 0.5f) * xis) / wxsum;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float pv = (float)img[(((size_t)b * 3 + c) * H + (ylo + j)) * W + xlo + i];
                    if (round16) pv = (float)(f16)pv;
                    h[c] += wx * pv;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) acc[c] += wy * h[c];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) out[(((size_t)b * 3 + c) * OH + oy) * OW + ox] = round16 ? (float)(f16)acc[c] : acc[c];
    }
}
template <typename TIn>
int launch_resize_bicubic_aa(const void* img, float* out, int B, int H, int W, int OH, int OW, int round16, hipStream_t st) {
    const long total = (long)B * OH * OW;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL((resize_bicubic_aa_kernel<TIn>), dim3(blocks), dim3(256), 0, st, (const TIn*)img, out, B, H, W, OH, OW, round16);
    return (int)hipGetLastError();
}
template int launch_resize_bicubic_aa<float>(const void*, float*, int, int, int, int, int, int, hipStream_t);
template int launch_resize_bicubic_aa<f16>(const void*, float*, int, int, int, int, int, int, hipStream_t);

// GroupNorm (v1.py:44,47: nn.GroupNorm(G, C), eps 1e-5) on an NHWC map, fused with the ReLU that follows it.  Three deterministic steps:
//   gn_partial: every block reduces a fixed slab of pixels to per-group (sum, sum of squares) in fp32 -> part[b][blk][G][2]
//   gn_finalize: one thread per (b, group) adds the slabs in order in fp64 -> (mean, rstd)
//   gn_apply: y = act((x - mean) * rstd * gamma[c] + beta[c])   (ReLU for MoGe-1 and the v2 defaults; modules.py:31-40 for the others)
// C % CH == 0 and C / CH divides 256 (C in {32, 64, 128, 256} for the released models).
constexpr int GN_PIX_PER_BLOCK = 2048;
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, float* __restrict__ part, int HW, int C, int G, int nblk) {
    constexpr int CH = TT<T>::CH;
    const int cpr = C / CH;                         // 16-byte chunks per pixel
    const int b = blockIdx.y, blk = blockIdx.x;
    const int chunk = threadIdx.x % cpr, prow = threadIdx.x / cpr, ppb = 256 / cpr;
    const int cg = C / G;                           // channels per group
    const long p0 = (long)blk * GN_PIX_PER_BLOCK;
    const long p1 = p0 + GN_PIX_PER_BLOCK < HW ? p0 + GN_PIX_PER_BLOCK : HW;
    float s1 = 0.f, s2 = 0.f;
    const T* base = x + (size_t)b * HW * C + chunk * CH;
    for (long p = p0 + prow; p < p1; p += ppb) {
        float v[CH];
        if constexpr (CH == 8) {
            const f16x8 h = *reinterpret_cast<const f16x8*>(base + (size_t)p * C);
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (float)h[i];
        } else {
            const f32x4 h = *reinterpret_cast<const f32x4*>(base + (size_t)p * C);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = h[i];
        }
#pragma unroll
        for (int i = 0; i < CH; i++) { s1 += v[i]; s2 = fmaf(v[i], v[i], s2); }
    }
    __shared__ float sh1[256], sh2[256];
    sh1[threadIdx.x] = s1; sh2[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < G) {                          // fixed-order sum of this group's threads (a chunk never straddles groups: cg % CH == 0)
        const int g = threadIdx.x;
        double a1 = 0.0, a2 = 0.0;
        for (int t = 0; t < 256; t++) {
            const int c0 = (t % cpr) * CH;
            if (c0 / cg == g) { a1 += sh1[t]; a2 += sh2[t]; }
        }
        float* o = part + (((size_t)b * nblk + blk) * G + g) * 2;
        o[0] = (float)a1; o[1] = (float)a2;
    }
}
__global__ void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ mr, int B, int G, int nblk, double count, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i - b * G;
    double a1 = 0.0, a2 = 0.0;
    for (int k = 0; k < nblk; k++) {
        const float* p = part + (((size_t)b * nblk + k) * G + g) * 2;
        a1 += p[0]; a2 += p[1];
    }
    const double mean = a1 / count;
    double var = a2 / count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    mr[2 * i] = (float)mean;
    mr[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
// activation of a residual block (modules.py:31-40): ReLU, LeakyReLU(0.2), SiLU, ELU(alpha = 1); codes = moge_activation
__device__ __forceinline__ float res_act(float v, int act) {
    switch (act) {
        case 1: return v > 0.f ? v : 0.2f * v;
        case 2: return v / (1.f + expf(-v));
        case 3: return v > 0.f ? v : expm1f(v);
        default: return fmaxf(v, 0.f);
    }
}
// y = act((x - mean[g]) * rstd[g] * gamma[c] + beta[c]); gamma / beta may be null (InstanceNorm2d has no affine part), mr may be null (no norm:
// the bare activation of a block whose norm is Identity and whose activation the conv kernels do not fuse).  cg = channels per group (1 = per channel).
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_act_kernel(const T* x, T* y /* may alias x: elementwise */, const float* __restrict__ mr,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, unsigned chunks_per_image, int C, int G, int act) {
    // grid (blocks, B): the image index comes from blockIdx.y and everything inside an image is 32-bit arithmetic (the first form of this kernel
    // spent three 64-bit divisions and sixteen 4-byte gathers per 16 bytes of data: 2.6 ms for a 655 MB map, profiles/r05x_v1_kernel_stats.csv)
    constexpr int CH = TT<T>::CH;
    const unsigned cpr = C / CH, cg = C / G;
    const unsigned b = blockIdx.y;
    const size_t img = (size_t)b * chunks_per_image * CH;
    const float* mrb = mr ? mr + 2 * (size_t)b * G : nullptr;
    // cpr divides 256 whenever there is a norm (launcher), so a thread keeps ONE channel chunk over its whole grid-stride walk: its statistics and affine
    // parameters are loaded once into registers (per-element gathers of gamma / beta ran the texture path at 17 instructions per 16 bytes of data)
    const bool fixed = (256u % cpr) == 0;
    const unsigned c0f = (threadIdx.x % cpr) * CH;
    float mean[CH], rstd[CH], ga[CH], be[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) { mean[i] = 0.f; rstd[i] = 1.f; ga[i] = 1.f; be[i] = 0.f; }
    if (mrb && fixed) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const unsigned g = (c0f + i) / cg;
            mean[i] = mrb[2 * g]; rstd[i] = mrb[2 * g + 1];
            if (gamma) { ga[i] = gamma[c0f + i]; be[i] = beta[c0f + i]; }
        }
    }
    // (four chunks per trip with the loads issued up front measured level: 12.3 ms of norm class either way - the in-place read + write stream, not latency)
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < chunks_per_image; idx += gridDim.x * 256u) {
        float v[CH];
        if constexpr (CH == 8) {
            const f16x8 h = *reinterpret_cast<const f16x8*>(x + img + (size_t)idx * CH);
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (float)h[i];
        } else {
            const f32x4 h = *reinterpret_cast<const f32x4*>(x + img + (size_t)idx * CH);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = h[i];
        }
#pragma unroll
        for (int i = 0; i < CH; i++) {
            float t = v[i];
            if (mrb) {                                       // (mrb != null implies `fixed`: launch_groupnorm_act)
                t = (t - mean[i]) * rstd[i];
                if (gamma) t = t * ga[i] + be[i];
            }
            v[i] = res_act(t, act);
        }
        if constexpr (CH == 8) {
            f16x8 h;
#pragma unroll
            for (int i = 0; i < 8; i++) h[i] = (f16)v[i];
            *reinterpret_cast<f16x8*>(y + img + (size_t)idx * CH) = h;
        } else {
            *reinterpret_cast<f32x4*>(y + img + (size_t)idx * CH) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}
// InstanceNorm2d statistics (modules.py:50,56: nn.InstanceNorm2d(C), eps 1e-5, no affine, instance statistics in eval mode too): the slab
// scheme of gn_partial with one (sum, sum of squares) pair per CHANNEL -> part[b][blk][C][2]; gn_finalize then runs with G = C.
template <typename T>
__global__ __launch_bounds__(256) void in_partial_kernel(const T* __restrict__ x, float* __restrict__ part, int HW, int C, int nblk) {
    constexpr int CH = TT<T>::CH;
    const int cpr = C / CH;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int chunk = threadIdx.x % cpr, prow = threadIdx.x / cpr, ppb = 256 / cpr;
    const long p0 = (long)blk * GN_PIX_PER_BLOCK;
    const long p1 = p0 + GN_PIX_PER_BLOCK < HW ? p0 + GN_PIX_PER_BLOCK : HW;
    float s1[CH], s2[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) { s1[i] = 0.f; s2[i] = 0.f; }
    const T* base = x + (size_t)b * HW * C + chunk * CH;
    for (long p = p0 + prow; p < p1; p += ppb) {
        float v[CH];
        if constexpr (CH == 8) {
            const f16x8 h = *reinterpret_cast<const f16x8*>(base + (size_t)p * C);
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (float)h[i];
        } else {
            const f32x4 h = *reinterpret_cast<const f32x4*>(base + (size_t)p * C);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = h[i];
        }
#pragma unroll
        for (int i = 0; i < CH; i++) { s1[i] += v[i]; s2[i] = fmaf(v[i], v[i], s2[i]); }
    }
    __shared__ float sh1[256 * CH], sh2[256 * CH];
#pragma unroll
    for (int i = 0; i < CH; i++) { sh1[threadIdx.x * CH + i] = s1[i]; sh2[threadIdx.x * CH + i] = s2[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {            // fixed-order sum over the pixel rows that hold channel c
        const int ck = c / CH, i = c - ck * CH;
        double a1 = 0.0, a2 = 0.0;
        for (int r = 0; r < ppb; r++) { a1 += sh1[(r * cpr + ck) * CH + i]; a2 += sh2[(r * cpr + ck) * CH + i]; }
        float* o = part + (((size_t)b * nblk + blk) * C + c) * 2;
        o[0] = (float)a1; o[1] = (float)a2;
    }
}
// scratch: part (B * nblk * G * 2 floats) followed by mr (B * G * 2 floats); groupnorm_scratch_floats(B, H, W, G) floats.
// G = 0: no normalisation (activation only, scratch unused); G = C with cg < a 16-byte chunk: per-channel statistics (InstanceNorm2d, pass null gamma / beta).
template <typename T>
int launch_groupnorm_act(const void* x, void* y, const float* gamma, const float* beta, float* scratch, int B, int H, int W, int C, int G, int act, hipStream_t st) {
    constexpr int CH = TT<T>::CH;
    if (C % CH || act < 0 || act > 3 || (G != 0 && 256 % (C / CH))) return -1;
    const long HW = (long)H * W;
    const long cpi = HW * (C / CH);                       // 16-byte chunks per image
    if (cpi >= (1L << 31) || B > 65535) return -1;
    long nb = (cpi + 255) / 256;
    const int per_image = (int)(nb > 4096 ? 4096 : nb);   // grid-stride inside an image
    const dim3 agrid(per_image, B);
    if (G == 0) {
        hipLaunchKernelGGL((gn_apply_act_kernel<T>), agrid, dim3(256), 0, st, (const T*)x, (T*)y, (const float*)nullptr, gamma, beta, (unsigned)cpi, C, 1, act);
        return (int)hipGetLastError();
    }
    if (C % G) return -1;
    const bool per_channel = G == C;
    if (!per_channel && ((C / G) % CH || G > 64)) return -1;
    const int nblk = (int)((HW + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK);
    float* part = scratch;
    float* mr = scratch + (size_t)B * nblk * G * 2;
    if (per_channel) hipLaunchKernelGGL((in_partial_kernel<T>), dim3(nblk, B), dim3(256), 0, st, (const T*)x, part, (int)HW, C, nblk);
    else hipLaunchKernelGGL((gn_partial_kernel<T>), dim3(nblk, B), dim3(256), 0, st, (const T*)x, part, (int)HW, C, G, nblk);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * G + 63) / 64), dim3(64), 0, st, part, mr, B, G, nblk, (double)HW * (C / G), 1e-5f);
    hipLaunchKernelGGL((gn_apply_act_kernel<T>), agrid, dim3(256), 0, st, (const T*)x, (T*)y, mr, gamma, beta, (unsigned)cpi, C, G, act);
    return (int)hipGetLastError();
}
template <typename T>
int launch_groupnorm_relu(const void* x, void* y, const float* gamma, const float* beta, float* scratch, int B, int H, int W, int C, int G, hipStream_t st) {
    if (G <= 0 || G == C) return -1;                       // (GroupNorm proper: MoGe-1's blocks and the "layer_norm" / "group_norm" options)
    return launch_groupnorm_act<T>(x, y, gamma, beta, scratch, B, H, W, C, G, 0, st);
}
size_t groupnorm_scratch_floats(int B, int H, int W, int G) {
    const long HW = (long)H * W;
    const int nblk = (int)((HW + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK);
    return (size_t)B * nblk * G * 2 + (size_t)B * G * 2;
}
template int launch_groupnorm_relu<f16>(const void*, void*, const float*, const float*, float*, int, int, int, int, int, hipStream_t);
template int launch_groupnorm_relu<float>(const void*, void*, const float*, const float*, float*, int, int, int, int, int, hipStream_t);
template int launch_groupnorm_act<f16>(const void*, void*, const float*, const float*, float*, int, int, int, int, int, int, hipStream_t);
template int launch_groupnorm_act<float>(const void*, void*, const float*, const float*, float*, int, int, int, int, int, int, hipStream_t);

// v1.py:127-130: bilinear (align_corners=False, no antialias) resize of the NHWC feature map (B, hs, ws, C) to (B, OH, OW, .) with the view-plane
// uv of the OUTPUT grid appended as channels C, C+1 (aspect = OW / OH) and zeros up to the padded pitch Cp (a multiple of 8: the 3x3 conv
// that follows reads 16-byte chunks)
template <typename T>
__global__ __launch_bounds__(256) void resize_bilinear_uv_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int hs, int ws, int C, int OH, int OW, int Cp,
                                                                 float u0, float u1, float ustep, float v0, float v1, float vstep) {
    constexpr int CH = TT<T>::CH;
    const int cpr = Cp / CH;
    const long total = (long)B * OH * OW * cpr;
    const float sy_scale = (float)hs / (float)OH, sx_scale = (float)ws / (float)OW;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int chunk = idx % cpr;
        long t = idx / cpr;
        const int ox = t % OW; t /= OW;
        const int oy = t % OH;
        const int b = t / OH;
        const int c0 = chunk * CH;
        float v[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) v[i] = 0.f;
        if (c0 < C) {
            float sy = sy_scale * (oy + 0.5f) - 0.5f, sx = sx_scale * (ox + 0.5f) - 0.5f;
            sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < hs - 1 ? 1 : 0), x1 = x0 + (x0 < ws - 1 ? 1 : 0);
            const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
            const T* base = x + (size_t)b * hs * ws * C + c0;
            float a[CH], bq[CH], cq[CH], dq[CH];
            load4(base + ((size_t)y0 * ws + x0) * C, a); load4(base + ((size_t)y0 * ws + x1) * C, bq);
            load4(base + ((size_t)y1 * ws + x0) * C, cq); load4(base + ((size_t)y1 * ws + x1) * C, dq);
            if constexpr (CH == 8) {
                load4(base + ((size_t)y0 * ws + x0) * C + 4, a + 4); load4(base + ((size_t)y0 * ws + x1) * C + 4, bq + 4);
                load4(base + ((size_t)y1 * ws + x0) * C + 4, cq + 4); load4(base + ((size_t)y1 * ws + x1) * C + 4, dq + 4);
            }
#pragma unroll
            for (int i = 0; i < CH; i++) v[i] = ly0 * (lx0 * a[i] + lx1 * bq[i]) + ly1 * (lx0 * cq[i] + lx1 * dq[i]);
        }
        if (c0 <= C && C < c0 + CH) {                 // the chunk that holds channels C, C + 1 (C % CH == 0: they start the chunk)
            v[C - c0] = linspace_at(u0, u1, ustep, OW, ox);
            v[C - c0 + 1] = linspace_at(v0, v1, vstep, OH, oy);
        }
        T* o = out + (((size_t)b * OH + oy) * OW + ox) * Cp + c0;
        store4(o, v[0], v[1], v[2], v[3]);
        if constexpr (CH == 8) store4(o + 4, v[4], v[5], v[6], v[7]);
    }
}
template <typename T>
int launch_resize_bilinear_uv(const void* x, void* out, int B, int hs, int ws, int C, int OH, int OW, int Cp, float u0, float u1, float v0, float v1, hipStream_t st) {
    if (C % TT<T>::CH || Cp % TT<T>::CH || Cp < C + 2) return -1;
    const long total = (long)B * OH * OW * (Cp / TT<T>::CH);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65536) blocks = 65536;
    const float ustep = OW > 1 ? (u1 - u0) / (float)(OW - 1) : 0.f, vstep = OH > 1 ? (v1 - v0) / (float)(OH - 1) : 0.f;
    hipLaunchKernelGGL((resize_bilinear_uv_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)x, (T*)out, B, hs, ws, C, OH, OW, Cp, u0, u1, ustep, v0, v1, vstep);
    return (int)hipGetLastError();
}
template int launch_resize_bilinear_uv<f16>(const void*, void*, int, int, int, int, int, int, int, float, float, float, float, hipStream_t);
template int launch_resize_bilinear_uv<float>(const void*, void*, int, int, int, int, int, int, int, float, float, float, float, hipStream_t);


// --------------------------------------------------------------------------------------------
// A-operand table of conv_pp.hip's fused output conv (GemmArgs::dot_tab): nd groups of four weight rows W[4*gq + e][0..31] (fp32, rows >=
// `rows` are zero) as v_mfma_f32_16x16x32_f16 A fragments: lane (r = lane & 15, g4 = lane >> 4) holds row (r & 3) of its group - replicated
// over the four row groups, so every lane group of the result holds the four outputs - at the K positions of its slots s = 0..7:
// channel 4*g4 + s (s < 4), 16 + 4*g4 + s - 4 (s >= 4) = the channel order of the conv accumulators that form the B operand.
// --------------------------------------------------------------------------------------------
__global__ void pack_dot_table_kernel(const float* __restrict__ w, int rows, int nd, f16* __restrict__ tab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // (gq, lane, s)
    if (idx >= nd * 64 * 8) return;
    const int s = idx & 7, lane = (idx >> 3) & 63, gq = idx >> 9;
    const int row = 4 * gq + (lane & 3), g4 = lane >> 4;
    const int c = s < 4 ? 4 * g4 + s : 16 + 4 * g4 + (s - 4);
    tab[idx] = (f16)(row < rows ? w[row * 32 + c] : 0.f);
}
int launch_pack_dot_table(const float* w, int rows, int nd, void* tab, hipStream_t st) {
    if (rows < 1 || nd < 1 || rows > 4 * nd) return -1;
    hipLaunchKernelGGL(pack_dot_table_kernel, dim3((nd * 512 + 255) / 256), dim3(256), 0, st, w, rows, nd, (f16*)tab);
    return (int)hipGetLastError();
}
