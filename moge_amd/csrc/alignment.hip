// Optimal-alignment solvers of the evaluation path (reference moge/utils/alignment.py, used by moge/test/metrics.py:128-282).
//
// The core problem (alignment.py:52-89, trunc=None):   a* = argmin_a  sum_i w_i |a x_i - y_i|      w_i >= 0
// is a weighted median: after flipping signs so that x_i >= 0, a* is one of the ratios r_i = y_i / max(x_i, eps), the first one (in ascending
// order) at which the derivative 2 * prefix(w x) - total(w x) stops being negative.  The reference sorts the ratios of every row with
// torch.sort, gathers, takes a cumsum and a searchsorted - for the affine solvers on an (anchors x 3n) matrix it materialises first
// (alignment.py:331-336: 4096 anchors x 12288 residuals = 600 MB of temporaries at the 64 x 64 evaluation grid).
//
// Here: ONE workgroup per row, the row never leaves the CU.
//   1. residuals are formed on the fly (anchored modes subtract the anchor sample while loading: nothing is materialised),
//      ratio + element index + w*x go to LDS (4 + 2 + 4 bytes per element: 16384 elements = 160 KiB is exactly the CU's LDS, so a row
//      holds up to ALIGN_MAX_N = 15360 elements; the evaluation grid needs 12288);
//   2. bitonic sort of (ratio, index) pairs in LDS, lexicographic = the order of a stable sort;
//   3. prefix sums of w*x in sorted order in FLOAT64 (block scan), first position with 2 * prefix - total >= 0;
//   4. objective value at the solution from a second pass over the inputs (float64 accumulation).
// HBM traffic = the inputs, twice; everything else is LDS.  Bound: LDS bandwidth of the sort (105 passes over 96 KiB for a 16384 sort).
// The truncated objective (alignment.py:91-144) belongs to the training losses and is not built (training is out of scope).
#include "common.h"
#include "../../include/moge_hip.h"

constexpr int ALIGN_THREADS = 1024;
constexpr int ALIGN_MAX_N = 15360;

struct AlignArgs {
    // plain rows: x, y, w are (rows, n)
    const float* x; const float* y; const float* w;
    // anchored rows: src / tgt (B, n, d), wt (B, n); row r solves batch row_b[r] with sample row_k[r] subtracted from the components in comp_mask
    const float* src; const float* tgt; const float* wt;
    const int* row_b; const int* row_k;
    int n, d, comp_mask;
    float eps;
    float* a; float* loss; int* index;
};

struct AlignRow {          // how this row's element j is formed
    const float* x; const float* y; const float* w;      // plain: row pointers;  anchored: batch pointers
    float ax[3], ay[3];
    int d;
    bool anchored;
};

__device__ __forceinline__ void align_fetch_signed(const AlignRow& r, int j, float& x, float& y, float& w) {
    if (!r.anchored) {
        x = r.x[j]; y = r.y[j]; w = r.w[j];
    } else {
        int i = j, c = 0;
        if (r.d == 3) { i = j / 3; c = j - 3 * i; }
        x = r.x[j] - r.ax[c];                              // alignment.py:191 / :274 / :331
        y = r.y[j] - r.ay[c];
        w = r.w[i];
    }
    const float s = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);   // :71-72 (sign(0) = 0: the element drops out with ratio 0, weight 0)
    x *= s; y *= s;
}

__device__ __forceinline__ double warp_incl_scan(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

__global__ __launch_bounds__(ALIGN_THREADS) void align_l1_kernel(const AlignArgs g, int NE, int NP) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* keys = reinterpret_cast<float*>(smem);                                     // [NP]
    float* wxs = reinterpret_cast<float*>(smem + (size_t)NP * 4);                     // [NE]
    unsigned short* idx = reinterpret_cast<unsigned short*>(smem + (size_t)NP * 4 + (size_t)NE * 4);      // [NP]
    __shared__ double wsum[16];
    __shared__ int found;
    __shared__ float sol_a;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    AlignRow r;
    r.anchored = g.src != nullptr;
    r.d = g.d;
    if (r.anchored) {
        const int b = g.row_b[row], k = g.row_k[row];
        r.x = g.src + (size_t)b * g.n * g.d;
        r.y = g.tgt + (size_t)b * g.n * g.d;
        r.w = g.wt + (size_t)b * g.n;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const bool on = c < g.d && ((g.comp_mask >> c) & 1) && k >= 0;
            r.ax[c] = on ? r.x[(size_t)k * g.d + c] : 0.f;
            r.ay[c] = on ? r.y[(size_t)k * g.d + c] : 0.f;
        }
    } else {
        r.x = g.x + (size_t)row * NE; r.y = g.y + (size_t)row * NE; r.w = g.w + (size_t)row * NE;
    }
    if (tid == 0) found = NE - 1;                                                      // :78 clamp_max(n - 1)

    // ---- 1. ratios, indices, w*x -> LDS ------------------------------------------------------------------------------------------
    const float inf = __builtin_inff();
    for (int j = tid; j < NP; j += ALIGN_THREADS) {
        float key = inf;
        if (j < NE) {
            float x, y, w;
            align_fetch_signed(r, j, x, y, w);
            key = y / fmaxf(x, g.eps);                                                 // :73
            if (key != key) key = inf;                                                 // NaN sorts last (torch.sort), like the padding
            wxs[j] = x * w;                                                            // :76
        }
        keys[j] = key;
        idx[j] = (unsigned short)j;
    }
    __syncthreads();

    // ---- 2. bitonic sort of (key, index), ascending ------------------------------------------------------------------------------
    for (int k = 2; k <= NP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < (NP >> 1); p += ALIGN_THREADS) {
                const int lo = ((p & ~(j - 1)) << 1) | (p & (j - 1)), hi = lo | j;     // the pair (lo, lo ^ j), lo has bit j clear
                const float ka = keys[lo], kb = keys[hi];
                const unsigned short ia = idx[lo], ib = idx[hi];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                const bool up = (lo & k) == 0;
                if (gt == up) { keys[lo] = kb; keys[hi] = ka; idx[lo] = ib; idx[hi] = ia; }
            }
            __syncthreads();
        }

    // ---- 3. prefix sums of w*x in sorted order (float64), first position whose derivative is >= 0 ------------------------------
    const int C = NP / ALIGN_THREADS > 0 ? NP / ALIGN_THREADS : 1;                     // consecutive elements per thread
    const int p0 = tid * C;
    double local = 0.0;
    for (int q = 0; q < C; q++) {
        const int p = p0 + q;
        if (p < NP) { const int e = idx[p]; if (e < NE) local += (double)wxs[e]; }
    }
    const double incl = warp_incl_scan(local, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    double base = 0.0, total = 0.0;
#pragma unroll
    for (int v = 0; v < 16; v++) { const double s = wsum[v]; if (v < wave) base += s; total += s; }
    double cum = base + incl - local;
    int first = 0x7fffffff;
    for (int q = 0; q < C; q++) {
        const int p = p0 + q;
        if (p < NE) {                                                                  // (sorted positions >= NE hold padding)
            const int e = idx[p];
            if (e < NE) cum += (double)wxs[e];
            if (first == 0x7fffffff && 2.0 * cum - total >= 0.0) first = p;            // :77-78
        }
    }
    if (first != 0x7fffffff) atomicMin(&found, first);
    __syncthreads();
    const int pos = found;
    if (tid == 0) {
        sol_a = keys[pos];                                                             // :80
        g.a[row] = keys[pos];
        g.index[row] = (int)idx[pos];                                                  // :81
    }
    __syncthreads();

    // ---- 4. objective value at the solution ------------------------------------------------------------------------------------------
    const float a = sol_a;
    double part = 0.0;
    for (int j = tid; j < NE; j += ALIGN_THREADS) {
        float x, y, w;
        align_fetch_signed(r, j, x, y, w);
        part += (double)(w * fabsf(a * x - y));                                        // :82
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    __syncthreads();
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int v = 0; v < 16; v++) s += wsum[v];
        g.loss[row] = (float)s;
    }
}

// per batch element: the minimum of loss over its rows and the LAST row attaining it (alignment.py:13-20 as the indexed assignment runs on CPU)
__global__ __launch_bounds__(256) void align_select_kernel(const float* loss, const int* row_b, int rows, float* min_loss, int* min_row) {
    __shared__ float smin[256];
    __shared__ int srow[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float best = __builtin_inff();
    int brow = -1;
    for (int r = tid; r < rows; r += 256)
        if (row_b[r] == b) {
            const float l = loss[r];
            if (brow < 0 || l <= best) { best = l; brow = r; }          // r ascends: a tie keeps the later row
        }
    smin[tid] = best; srow[tid] = brow;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float l = smin[tid + o];
            const int r = srow[tid + o];
            if (r >= 0 && (srow[tid] < 0 || l < smin[tid] || (l == smin[tid] && r > srow[tid]))) { smin[tid] = l; srow[tid] = r; }
        }
        __syncthreads();
    }
    if (tid == 0) { min_loss[b] = smin[0]; min_row[b] = srow[0]; }
}

// alignment.py:399-415: min sum_i (sqrt(w_i) x_i a + b - sqrt(w_i) y_i)^2 per row (the constant column is not weighted in the reference);
// normal equations, float64 sums
__global__ __launch_bounds__(1024) void align_lstsq_kernel(const float* x, const float* y, const float* w, int n, float* a, float* b) {
    __shared__ double red[16][4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = x + (size_t)row * n;
    const float* yr = y + (size_t)row * n;
    const float* wr = w ? w + (size_t)row * n : nullptr;
    double s[4] = {0.0, 0.0, 0.0, 0.0};          // sum u^2, sum u, sum u v, sum v
    for (int i = tid; i < n; i += 1024) {
        const float ws = wr ? sqrtf(wr[i]) : 1.f;
        const double u = (double)(ws * xr[i]), v = (double)(ws * yr[i]);
        s[0] += u * u; s[1] += u; s[2] += u * v; s[3] += v;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o);
        if (lane == 0) red[wave][q] = s[q];
    }
    __syncthreads();
    if (tid == 0) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int v = 0; v < 16; v++)
            for (int q = 0; q < 4; q++) t[q] += red[v][q];
        const double det = t[0] * (double)n - t[1] * t[1];
        a[row] = (float)((t[2] * (double)n - t[1] * t[3]) / det);
        b[row] = (float)((t[0] * t[3] - t[1] * t[2]) / det);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// C ABI (include/moge_hip.h)
// ------------------------------------------------------------------------------------------------------------------------
static int align_launch(const AlignArgs& g, int rows, int NE, hipStream_t st) {
    if (rows <= 0) return 0;
    if (NE < 1 || NE > ALIGN_MAX_N) {
        moge_internal_set_error("moge_align_l1: a row holds 1 .. 15360 residuals (the row is sorted inside one CU's LDS)");
        return MOGE_ERR_INVALID;
    }
    int NP = ALIGN_THREADS;                              // at least one element per thread keeps the scan simple
    while (NP < NE) NP <<= 1;
    const size_t smem = (size_t)NP * 4 + (size_t)NE * 4 + (size_t)NP * 2;
    if (int rc = set_dyn_lds<align_l1_kernel>((int)smem)) { moge_internal_set_error("moge_align_l1: cannot reserve LDS"); return MOGE_ERR_HIP; }
    hipLaunchKernelGGL(align_l1_kernel, dim3((unsigned)rows), dim3(ALIGN_THREADS), smem, st, g, NE, NP);
    if (hipGetLastError() != hipSuccess) { moge_internal_set_error("moge_align_l1: launch failed"); return MOGE_ERR_HIP; }
    return 0;
}

extern "C" {

int moge_align_l1(const float* x, const float* y, const float* w, int rows, int n, float eps, float* a, float* loss, int32_t* index, void* stream) {
    if (!x || !y || !w || !a || !loss || !index) { moge_internal_set_error("moge_align_l1: null argument"); return MOGE_ERR_INVALID; }
    AlignArgs g{};
    g.x = x; g.y = y; g.w = w; g.n = n; g.d = 1; g.eps = eps; g.a = a; g.loss = loss; g.index = index;
    return align_launch(g, rows, n, (hipStream_t)stream);
}

int moge_align_l1_anchored(const float* src, const float* tgt, const float* weight, int n, int d, int comp_mask, const int32_t* row_batch,
                           const int32_t* row_anchor, int rows, float eps, float* scale, float* loss, int32_t* index, void* stream) {
    if (!src || !tgt || !weight || !row_batch || !row_anchor || !scale || !loss || !index) { moge_internal_set_error("moge_align_l1_anchored: null argument"); return MOGE_ERR_INVALID; }
    if (d != 1 && d != 3) { moge_internal_set_error("moge_align_l1_anchored: d must be 1 (depth) or 3 (points)"); return MOGE_ERR_INVALID; }
    AlignArgs g{};
    g.src = src; g.tgt = tgt; g.wt = weight; g.row_b = row_batch; g.row_k = row_anchor;
    g.n = n; g.d = d; g.comp_mask = comp_mask; g.eps = eps; g.a = scale; g.loss = loss; g.index = index;
    return align_launch(g, rows, n * d, (hipStream_t)stream);
}

int moge_align_select(const float* loss, const int32_t* row_batch, int rows, int batch, float* min_loss, int32_t* min_row, void* stream) {
    if (!loss || !row_batch || !min_loss || !min_row) { moge_internal_set_error("moge_align_select: null argument"); return MOGE_ERR_INVALID; }
    if (batch <= 0) return 0;
    hipLaunchKernelGGL(align_select_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, loss, row_batch, rows, min_loss, min_row);
    return hipGetLastError() == hipSuccess ? 0 : MOGE_ERR_HIP;
}

int moge_align_lstsq(const float* x, const float* y, const float* w, int rows, int n, float* a, float* b, void* stream) {
    if (!x || !y || !a || !b || n < 2) { moge_internal_set_error("moge_align_lstsq: null argument or fewer than two samples"); return MOGE_ERR_INVALID; }
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(align_lstsq_kernel, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, x, y, w, n, a, b);
    return hipGetLastError() == hipSuccess ? 0 : MOGE_ERR_HIP;
}

}   // extern "C"
