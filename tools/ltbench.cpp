// Vendor-library reference point: hipBLASLt fp16 TN GEMM (fp32 accumulate) on the encoder's GEMM shapes, same device,
// random data.  Not part of the product; used to place gemm_pp128's numbers.
// build: hipcc --offload-arch=gfx950 -O2 tools/ltbench.cpp -o tools/ltbench -lhipblaslt
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill(__half* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = __float2half(((h & 0xffff) / 65536.f - 0.5f) * 2.f);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 10;
    struct S { const char* name; int64_t M, N, K; int gelu; } shapes[] = {
        {"qkv", 115232, 3072, 1024, 0}, {"proj", 115232, 1024, 1024, 0}, {"fc1", 115232, 4096, 1024, 0}, {"fc1+gelu_bias", 115232, 4096, 1024, 1},
        {"fc2", 115232, 1024, 4096, 0}, {"sq8192", 8192, 8192, 8192, 0}};
    hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
    hipStream_t st; CK(hipStreamCreate(&st));
    size_t wsz = 256u << 20; void* ws; CK(hipMalloc(&ws, wsz));
    for (auto& s : shapes) {
        __half *A, *W, *C, *bias;
        CK(hipMalloc(&A, s.M * s.K * 2)); CK(hipMalloc(&W, s.N * s.K * 2)); CK(hipMalloc(&C, s.M * s.N * 2)); CK(hipMalloc(&bias, s.N * 2));
        fill<<<2048, 256, 0, st>>>(A, s.M * s.K, 1); fill<<<2048, 256, 0, st>>>(W, s.N * s.K, 2); fill<<<64, 256, 0, st>>>(bias, s.N, 3);
        hipblasLtMatmulDesc_t d; CK(hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t T = HIPBLAS_OP_T, Nn = HIPBLAS_OP_N;
        CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &T, sizeof(T)));
        CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &Nn, sizeof(Nn)));
        if (s.gelu) {
            hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_GELU_BIAS;
            CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
            CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
        }
        hipblasLtMatrixLayout_t la, lb, lc;
        CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, s.K, s.N, s.K));   // W as K x N col-major, transposed
        CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, s.K, s.M, s.K));   // A as K x M col-major
        CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16F, s.N, s.M, s.N));   // C^T
        hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
        hipblasLtMatmulHeuristicResult_t res[8]; int n = 0;
        CK(hipblasLtMatmulAlgoGetHeuristic(h, d, la, lb, lc, lc, pref, 8, res, &n));
        float alpha = 1.f, beta = 0.f;
        double best = 1e30;
        for (int a = 0; a < n; a++) {
            if (hipblasLtMatmul(h, d, &alpha, W, la, A, lb, &beta, C, lc, C, lc, &res[a].algo, ws, wsz, st) != 0) continue;
            CK(hipStreamSynchronize(st));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; i++) CK(hipblasLtMatmul(h, d, &alpha, W, la, A, lb, &beta, C, lc, C, lc, &res[a].algo, ws, wsz, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
            printf("hipblaslt %-14s M=%lld N=%lld K=%lld algo %d/%d  %.3f ms  %.1f TF/s\n", s.name, (long long)s.M, (long long)s.N, (long long)s.K, a, n, ms,
                   2.0 * s.M * s.N * s.K / (ms * 1e-3) / 1e12);
            fflush(stdout);
            if (ms < best) best = ms;
        }
        printf("hipblaslt %-14s BEST %.3f ms  %.1f TF/s\n", s.name, best, 2.0 * s.M * s.N * s.K / (best * 1e-3) / 1e12);
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(bias));
    }
    return 0;
}
