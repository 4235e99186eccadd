// Store-rate probe: what does a CU sustain when its 8 waves write a 256 x 256 fp16 output tile (128 KiB) the way the GEMM epilogues of this
// library do - 16 buffer_store_dwordx4 per wave, each covering 8 rows x 128 B of a row-major matrix - and what changes it: the lane -> address
// pattern, the cache-policy bits, how many CUs store at once, stores back to back or in one burst per tile period?
//   hipcc --offload-arch=gfx950 -O2 tools/store_rate.cpp -o tools/store_rate
// One workgroup per CU (160 KiB of LDS claimed), 8 waves.  A "tile" = every storing wave issues 16 stores (its 128 rows x 128 B); the destination
// advances by 256 rows per tile (a column of tiles of an M x N fp16 matrix, N = 1024 or 4096), so nothing is rewritten inside one launch.
//   rate   = bytes / time of launches that store back to back
//   burst  = s_memtime clocks from the first store to the last store ISSUED (what an epilogue waits for) and to vmcnt(0) (acknowledged),
//            with `gap` microseconds of s_sleep between the tiles (the GEMM's main loop)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// PAT 0: 8 rows x 128 B per instruction (the epilogues' pattern)   1: 1 KiB contiguous   2: 16 rows x 64 B   3: 64 rows x 16 B (row per lane)
//     4: 8 rows x 128 B as TWO dwordx2 per 16 bytes (twice the instructions, half the width)
template <int PAT, int AUX>
__global__ __launch_bounds__(512) void st_kernel(char* dst, unsigned ldc, unsigned long long tile_stride, int tiles, int waves_storing, int gap_sleeps,
                                                 unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                    // 2 x 4 waves: 128 rows x 64 fp16 columns each
    char* base = dst + (size_t)blockIdx.x * 512 + (size_t)wm * 128 * ldc + (size_t)wn * 128;       // the workgroup's 256 x 256 tile column
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    unsigned off[16];
#pragma unroll
    for (int it = 0; it < 16; it++) {
        if (PAT == 0 || PAT == 4) off[it] = (unsigned)(it * 8 + (lane >> 3)) * ldc + (unsigned)(lane & 7) * 16u;
        if (PAT == 1) off[it] = (unsigned)(it * 8) * ldc + (unsigned)lane * 16u;                       // (not a matrix layout: 1 KiB runs)
        if (PAT == 2) off[it] = (unsigned)((it >> 1) * 16 + (lane >> 2)) * ldc + (unsigned)(it & 1) * 64u + (unsigned)(lane & 3) * 16u;
        if (PAT == 3) off[it] = (unsigned)((it >> 3) * 64 + lane) * ldc + (unsigned)(it & 7) * 16u;
    }
    u32x4 v = {(unsigned)tid, 2u, 3u, 4u};
    unsigned long long t_issue = 0, t_ack = 0;
    unsigned soff = 0;
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; t++) {
        __builtin_amdgcn_s_barrier();
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        if (wave < waves_storing) {
#pragma unroll
            for (int it = 0; it < 16; it++) {
                if (PAT == 4) {
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{v[0], v[1]}, rs, (int)off[it], (int)soff, AUX);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{v[2], v[3]}, rs, (int)off[it] + 8, (int)soff, AUX);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off[it], (int)soff, AUX);
                }
            }
        }
        const unsigned long long b = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long c = __builtin_amdgcn_s_memtime();
        t_issue += b - a; t_ack += c - a;
        soff += (unsigned)tile_stride;
        for (int s = 0; s < gap_sleeps; s++) __builtin_amdgcn_s_sleep(127);       // 127 x 64 clocks
    }
    const unsigned long long w1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
    if (smem[tid] == 123) dst[0] = 1;
    if (blockIdx.x == gridDim.x / 2 && lane == 0 && (wave == 0 || wave == 7)) {
        unsigned long long* o = clk + (wave ? 4 : 0);
        o[0] = t_issue; o[1] = t_ack; o[2] = c1 - c0; o[3] = w1 - w0;
    }
}

template <int PAT, int AUX>
static void run(const char* what, char* dst, size_t bytes, unsigned ldc, int grid, int waves, int gap_us, unsigned long long* clk) {
    const int tiles = gap_us ? 12 : 24;
    const unsigned long long tile_stride = 256ull * ldc;
    if ((size_t)grid * 512 > ldc && ldc != 0) { /* more tile columns than the row holds: wrap rows by the caller's choice of ldc */ }
    if (tile_stride * tiles + 512ull * grid > bytes || tile_stride * tiles > 0xffffffffull) { printf("%s: does not fit\n", what); return; }
    CK(hipFuncSetAttribute((const void*)st_kernel<PAT, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int sleeps = gap_us ? (int)(gap_us * 1950.0 / (127 * 64)) : 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    st_kernel<PAT, AUX><<<grid, 512, 160 * 1024>>>(dst, ldc, tile_stride, tiles, waves, sleeps, clk); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) st_kernel<PAT, AUX><<<grid, 512, 160 * 1024>>>(dst, ldc, tile_stride, tiles, waves, sleeps, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    unsigned long long h[8]; CK(hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost));
    const double ghz = 0.1 * (double)h[2] / (double)h[3];
    const double per_tile = waves * 16.0 * 1024.0;              // bytes per CU and tile
    if (!gap_us)
        printf("%-34s grid %3d, %d waves: %6.1f B/clk/CU  %6.2f TB/s chip   (%.0f clocks per store instruction and CU, %.2f GHz)\n", what, grid, waves,
               per_tile * tiles * grid / (ms * 1e-3) / grid / (ghz * 1e9), per_tile * tiles * grid / (ms * 1e-3) / 1e12,
               (ms * 1e-3) * ghz * 1e9 / (tiles * waves * 16.0 * (PAT == 4 ? 2 : 1)), ghz);
    else
        printf("%-34s grid %3d, %d waves, %2d us between bursts: wave 0 issued after %5.0f clocks, acknowledged %5.0f;  wave 7 issued %5.0f, acknowledged %5.0f   (%.1f B/clk/CU over the burst)\n",
               what, grid, waves, gap_us, (double)h[0] / tiles, (double)h[1] / tiles, (double)h[4] / tiles, (double)h[5] / tiles, per_tile / ((double)h[5] / tiles));
    fflush(stdout);
}

int main() {
    const size_t bytes = 3ull << 30;
    char* dst; CK(hipMalloc(&dst, bytes)); CK(hipMemset(dst, 0, bytes));
    unsigned long long* clk; CK(hipMalloc(&clk, 64));
    const unsigned L1 = 2048, L4 = 8192;                        // fp16 rows of N = 1024 / 4096 (grid 256 x 512 B = 128 KiB of columns: rows wrap - tiles of several
                                                                // tile columns interleave as in the GEMM's grouped order; only the access pattern matters here)
    const unsigned LW = 256 * 512;                              // one row holds every workgroup's 512-byte tile row: distinct addresses chip-wide
    printf("-- back to back, pattern x cache policy (row pitch %u B)\n", LW);
    run<0, 0>("8 rows x 128 B", dst, bytes, LW, 256, 8, 0, clk);
    run<0, 2>("8 rows x 128 B, nt", dst, bytes, LW, 256, 8, 0, clk);
    run<0, 1>("8 rows x 128 B, sc0", dst, bytes, LW, 256, 8, 0, clk);
    run<0, 17>("8 rows x 128 B, sc0 sc1", dst, bytes, LW, 256, 8, 0, clk);
    run<0, 19>("8 rows x 128 B, sc0 sc1 nt", dst, bytes, LW, 256, 8, 0, clk);
    run<1, 0>("1 KiB contiguous", dst, bytes, LW, 256, 8, 0, clk);
    run<2, 0>("16 rows x 64 B", dst, bytes, LW, 256, 8, 0, clk);
    run<3, 0>("64 rows x 16 B (row per lane)", dst, bytes, LW, 256, 8, 0, clk);
    run<4, 0>("8 rows x 128 B as dwordx2 pairs", dst, bytes, LW, 256, 8, 0, clk);
    printf("-- back to back, fewer waves / fewer CUs\n");
    run<0, 0>("8 rows x 128 B", dst, bytes, LW, 256, 4, 0, clk);
    run<0, 0>("8 rows x 128 B", dst, bytes, LW, 256, 2, 0, clk);
    run<0, 0>("8 rows x 128 B", dst, bytes, LW, 256, 1, 0, clk);
    run<0, 0>("8 rows x 128 B", dst, bytes, LW, 64, 8, 0, clk);
    run<0, 0>("8 rows x 128 B", dst, bytes, LW, 8, 8, 0, clk);
    run<0, 0>("8 rows x 128 B", dst, bytes, LW, 1, 8, 0, clk);
    run<0, 2>("8 rows x 128 B, nt", dst, bytes, LW, 8, 8, 0, clk);
    printf("-- one 128 KiB burst per tile period (every CU at once, as the lock-step persistent GEMM does)\n");
    run<0, 0>("8 rows x 128 B", dst, bytes, LW, 256, 8, 25, clk);
    run<0, 2>("8 rows x 128 B, nt", dst, bytes, LW, 256, 8, 25, clk);
    run<0, 17>("8 rows x 128 B, sc0 sc1", dst, bytes, LW, 256, 8, 25, clk);
    run<1, 0>("1 KiB contiguous", dst, bytes, LW, 256, 8, 25, clk);
    run<0, 0>("8 rows x 128 B", dst, bytes, LW, 32, 8, 25, clk);
    run<0, 0>("8 rows x 128 B", dst, bytes, LW, 8, 8, 25, clk);
    (void)L1; (void)L4;
    return 0;
}
