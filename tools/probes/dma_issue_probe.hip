// What does one LDS-DMA piece (global_load_lds_dwordx4, 1 KiB per wave-instruction) cost a SIMD that is issuing back-to-back MFMAs - and does the SOURCE
// PATTERN matter?  The persistent GEMM issues 16 pieces per SIMD and K-tile (8 rows x 128 B each, row pitch = lda x 2 bytes); tools/probes/issue_probe says
// everything that is not an MFMA adds to the SIMD's time.  hipcc --offload-arch=gfx950 -O2 tools/probes/dma_issue_probe.hip -o tools/probes/dma_issue_probe
//
// Per iteration: 32 v_mfma_f32_16x16x32_f16 (compiler-visible, independent accumulators) + NP pieces, a vmcnt(NP) wait one iteration behind.
//   pattern 0: 8 rows x 128 B, row pitch 2048 B (the GEMM's K = 1024 A / W panels)     pattern 1: 1 KiB contiguous (a K-tile-blocked layout)
//   pattern 2: as 0 with the per-lane 64-bit address form (v_lshl_add_u64 + global_load_lds v[a:a+1], off)
// SPLIT: waves 0-3 issue only the MFMAs, waves 4-7 only the pieces (the ping-pong GEMM's arrangement, without its barriers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int NP, int PAT, int SPLIT>
__global__ __launch_bounds__(512, 2) void probe(const char* __restrict__ src, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[32];
    for (int i = 0; i < 32; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane * 3 + i)); }
    // a 2 MiB window per workgroup (L2-resident after the first pass), walked in 64 KiB steps
    const char* base = src + (size_t)(blockIdx.x & 63) * (2u << 20);
    unsigned off;
    if (PAT == 1) off = lane * 16;                                            // 1 KiB contiguous
    else off = (lane >> 3) * 2048 + (lane & 7) * 16;                          // 8 rows x 128 B, pitch 2048
    const bool do_mfma = !SPLIT || wave < 4, do_dma = !SPLIT || wave >= 4;
    for (int it = 0; it < iters; it++) {
        const char* p = base + (size_t)((it & 31) * 65536) + wave * (PAT == 1 ? 1024 * NP : 16384 * NP);
        const char* pu = (const char*)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)p >> 32)) << 32) |
                                       __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)p));
        if (do_dma) {
#pragma unroll
            for (int k = 0; k < NP; k++) {
                char* dst = smem + ((it & 1) * 8 + wave) * 8192 + k * 1024;
                if (PAT == 2) {
                    const char* pl = p + off + k * 16384;      // per-lane 64-bit pointer
                    __builtin_amdgcn_global_load_lds(GPTR(pl), LPTR(dst), 16, 0, 0);
                } else {
                    unsigned o = off + (PAT == 1 ? k * 1024 : k * 16384);
                    asm volatile("" : "+v"(o));
                    __builtin_amdgcn_global_load_lds(GPTR(pu + o), LPTR(dst), 16, 0, 0);
                }
            }
        }
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 32; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        if (do_dma && NP > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 32; i++) s += acc[i][0];
    if (sink && s == 12345.678f) sink[0] = s;
}
template <int NP, int PAT, int SPLIT>
static float run(const char* src, int threads, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)probe<NP, PAT, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    probe<NP, PAT, SPLIT><<<256, threads, 131072>>>(src, iters / 10 + 1, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    probe<NP, PAT, SPLIT><<<256, threads, 131072>>>(src, iters, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e6f / iters;       // ns per iteration
}
int main() {
    char* src; CK(hipMalloc(&src, (size_t)256 << 20)); CK(hipMemset(src, 0, (size_t)256 << 20));
    const int it = 20000;
    for (int rep = 0; rep < 2; rep++) {
        const float m1 = run<0, 0, 0>(src, 256, it), m2 = run<0, 0, 0>(src, 512, it);
        printf("32 MFMA per iteration, no DMA:            1 wave/SIMD %7.1f ns   2 waves/SIMD (2x the MFMAs) %7.1f ns\n", m1, m2);
        struct R { const char* n; float a, b, c; };
        R r[] = {
            {"4 pieces, 8 rows x 128 B pitch 2048, SADDR ", run<4, 0, 0>(src, 256, it), run<4, 0, 0>(src, 512, it), run<4, 0, 1>(src, 512, it)},
            {"4 pieces, 1 KiB contiguous, SADDR          ", run<4, 1, 0>(src, 256, it), run<4, 1, 0>(src, 512, it), run<4, 1, 1>(src, 512, it)},
            {"4 pieces, 8 rows x 128 B, 64-bit VGPR addr ", run<4, 2, 0>(src, 256, it), run<4, 2, 0>(src, 512, it), run<4, 2, 1>(src, 512, it)},
            {"8 pieces, 8 rows x 128 B pitch 2048, SADDR ", run<8, 0, 0>(src, 256, it), run<8, 0, 0>(src, 512, it), run<8, 0, 1>(src, 512, it)},
            {"8 pieces, 1 KiB contiguous, SADDR          ", run<8, 1, 0>(src, 256, it), run<8, 1, 0>(src, 512, it), run<8, 1, 1>(src, 512, it)},
        };
        for (auto& x : r)
            printf("%s  1 wave/SIMD %7.1f ns (+%5.1f per piece)   2 waves/SIMD each both %7.1f ns (+%5.1f)   split MFMA | DMA %7.1f ns (+%5.1f per piece over 1-wave MFMA)\n", x.n,
                   x.a, (x.a - m1) / (x.n[0] - '0'), x.b, (x.b - m2) / (2 * (x.n[0] - '0')), x.c, (x.c - m1) / (x.n[0] - '0'));
        printf("\n");
    }
    return 0;
}
