// Power/clock micro-benchmark: sustained TF/s and shader clock of bare MFMA chains (random fp16 operands in registers),
// v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16, 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O2 tools/mfma_power.cpp -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512, 2) void chain(const f16x8* __restrict__ in, float* out, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = in[(i * 64 + lane) % 4096]; b[i] = in[(1024 + i * 64 + lane) % 4096]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    float s = 0.f;
    if constexpr (KIND == 0) {
        f32x16 acc[8];
        for (int i = 0; i < 8; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + k) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][15];
    } else {
        f32x4 acc[32];
        for (int i = 0; i < 32; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int i = 0; i < 32; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + k) & 3], b[(i >> 3) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 32; i++) s += acc[i][0] + acc[i][3];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (s == 12345.f) out[0] = s;
    if (blockIdx.x == 100 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main() {
    f16x8* in; float* out; unsigned long long* clk;
    CK(hipMalloc(&in, 4096 * 16)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 16));
    _Float16* h = (_Float16*)malloc(4096 * 16);
    for (int i = 0; i < 4096 * 8; i++) h[i] = (_Float16)(((rand() & 0xffff) / 32768.f - 1.f));
    CK(hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice));
    const double secs = getenv("MFMA_POWER_SECONDS") ? atof(getenv("MFMA_POWER_SECONDS")) : 0.0;       // > 0: hold every configuration that long (SMI power sampling)
    // MFMA_POWER_ONLY="kind,zero,threads" (e.g. "1,0,512": 16x16x32, random operands, 8 waves per workgroup): run that one configuration (bench.py's live
    // `roofline.sustained_peak`)
    int only_kind = -1, only_zero = -1, only_wpb = -1;
    if (const char* o = getenv("MFMA_POWER_ONLY")) sscanf(o, "%d,%d,%d", &only_kind, &only_zero, &only_wpb);
    for (int zero = 0; zero < 2; zero++) {
        if (zero) CK(hipMemset(in, 0, 4096 * 16));
        for (int kind = 0; kind < 2; kind++)
            for (int wpb = 256; wpb <= 512; wpb *= 2) {
                if (only_kind >= 0 && (kind != only_kind || zero != only_zero || wpb != only_wpb)) continue;
                const int iters = 4000;
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                auto launch = [&]() { if (kind == 0) chain<0><<<256 * 4, wpb>>>(in, out, iters, clk); else chain<1><<<256 * 4, wpb>>>(in, out, iters, clk); };
                launch(); CK(hipDeviceSynchronize());
                int reps = 5;
                if (secs > 0) {                                   // one timed launch sizes the repeat count
                    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float one; CK(hipEventElapsedTime(&one, e0, e1));
                    reps = (int)(secs * 1e3 / one) + 1;
                    printf("  -> holding this configuration for %.1f s (%d launches)\n", secs, reps); fflush(stdout);
                }
                CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
                unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
                const double flops = 256.0 * 4 * (wpb / 64) * (double)iters * (kind == 0 ? 32 * 32768.0 : 64 * 16384.0);
                printf("%s data, %s, %d waves/WG (x4 WG/CU-slots): %.1f TF/s, shader clock %.0f MHz, cycles/MFMA/SIMD %.1f\n", zero ? "zero  " : "random", kind == 0 ? "32x32x16" : "16x16x32",
                       wpb / 64, flops / (ms * 1e-3) / 1e12, 100.0 * hc[0] / hc[1], (double)hc[0] / (iters * (kind == 0 ? 32.0 : 64.0)));
                fflush(stdout);
            }
    }
    return 0;
}
