// LDS-DMA fill-rate probe: how fast can a CU pull L2-resident data into LDS with global_load_lds_dwordx4, the way the GEMM / conv / attention
// kernels of this library stage their operand tiles (8 rows x 128 B per wave-instruction, optional XOR chunk swizzle on the SOURCE address)?
//   hipcc --offload-arch=gfx950 -O2 tools/lds_fill.cpp -o tools/lds_fill
// One workgroup per CU (grid 256 x {4, 8} waves, 128 KiB LDS ring); every wave issues pieces of 1 KiB and keeps `DEPTH` of them in flight
// (counted vmcnt).  Source: a per-workgroup window of `win` bytes (64 KiB: L2 hits after the first sweep; 8 MiB per workgroup = 2 GiB in all: streams
// through the Infinity Cache / HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int DEPTH, int SWZ>
__global__ __launch_bounds__(1024) void fill(const char* __restrict__ src, size_t win, int iters, unsigned long long* clk, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * win;
    const int prow = lane >> 3;
    const int chunk = SWZ ? ((lane & 7) ^ ((prow >> 1) & 7) ^ (wave & 7)) : (lane & 7);
    const unsigned off = (unsigned)(prow * 128 + chunk * 16);
    const unsigned pmask = (unsigned)(win / 1024) - 1;     // 1 KiB pieces in the window (a power of two): all index arithmetic is 32-bit scalar
    const unsigned long long bv = (unsigned long long)base;
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bv), bhi = __builtin_amdgcn_readfirstlane((unsigned)(bv >> 32));
    const char* ubase = (const char*)(((unsigned long long)bhi << 32) | blo);      // wave-uniform base in SGPRs
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    unsigned p = (unsigned)wave;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            const char* gu = ubase + (size_t)((p & pmask) << 10);
            __builtin_amdgcn_global_load_lds(GPTR(gu + off), LPTR(smem + (((wave * DEPTH * 2 + (it & 1) * DEPTH + k) & 127) << 10)), 16, 0, 0);
            p += (unsigned)nw;
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");       // the previous batch has landed, this one stays in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (sink && smem[tid * 16] == 123) sink[0] = 1.f;
    if (blockIdx.x == 100 && tid == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int DEPTH, int SWZ>
static void run(const char* src, size_t win, int threads, const char* what, unsigned long long* clk, float* sink) {
    const int iters = 2000;
    CK(hipFuncSetAttribute((const void*)fill<DEPTH, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fill<DEPTH, SWZ><<<256, threads, 131072>>>(src, win, iters, clk, sink); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) fill<DEPTH, SWZ><<<256, threads, 131072>>>(src, win, iters, clk, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
    const double bytes = 256.0 * (threads / 64) * (double)iters * DEPTH * 1024.0;
    const double ghz = 0.1 * (double)hc[0] / (double)hc[1];
    printf("%-44s %d waves, %d pieces in flight per wave, swizzle %d: %7.1f GB/s per CU, %6.2f TB/s chip, %5.1f B/clk/CU at %.2f GHz\n", what, threads / 64, DEPTH, SWZ,
           bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / (ghz * 1e9), ghz);
    fflush(stdout);
}

int main() {
    char* src; unsigned long long* clk; float* sink;
    const size_t big = (size_t)256 * (8u << 20);
    CK(hipMalloc(&src, big)); CK(hipMemset(src, 1, big)); CK(hipMalloc(&clk, 16)); CK(hipMalloc(&sink, 4));
    for (int threads = 64; threads <= 1024; threads *= 2) {
        run<4, 0>(src, 65536, threads, "64 KiB window per CU (L2 hits)", clk, sink);
        run<4, 1>(src, 65536, threads, "64 KiB window per CU (L2 hits)", clk, sink);
        run<8, 1>(src, 65536, threads, "64 KiB window per CU (L2 hits)", clk, sink);
        run<16, 1>(src, 65536, threads, "64 KiB window per CU (L2 hits)", clk, sink);
        run<8, 1>(src, (size_t)1 << 20, threads, "1 MiB window per CU (256 MiB: L2 misses, Infinity Cache)", clk, sink);
        run<8, 1>(src, (size_t)8 << 20, threads, "8 MiB window per CU (2 GiB: fabric / HBM)", clk, sink);
    }
    return 0;
}
