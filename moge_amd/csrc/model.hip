// Host side of libmoge_hip.so: handle, weight table + kernel-layout packing, workspace plan, the forward / infer
// drivers and the C ABI declared in include/moge_hip.h.  No device code here; every launch goes through launchers.h.
//
// Reference call structure reproduced (paths relative to the reference checkout):
//   forward   moge/model/v2.py:138-192   encoder moge/model/modules.py:120-136   ViT dinov2/models/vision_transformer.py:223-333
//   decoder   moge/model/modules.py:242-254 (ConvStack.forward)                  infer  moge/model/v2.py:194-303
#include "launchers.h"
#include "../../include/moge_hip.h"

#include <dlfcn.h>
#include <cmath>
#include <type_traits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
void moge_internal_set_error(const char* msg) { g_err = msg; }     // for the stateless entry points outside this file (alignment.hip)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MOGE_ERR_HIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define LCHK(x) do { int e_ = (x); if (e_ != 0) return fail(e_ < 0 ? MOGE_ERR_INVALID : MOGE_ERR_HIP, "%s: launch failed (%d: %s) (%s:%d)", #x, e_, e_ > 0 ? hipGetErrorString((hipError_t)e_) : "unsupported shape", __FILE__, __LINE__); } while (0)
#define CHK(x) do { int e_ = (x); if (e_ != 0) return e_; } while (0)

static const char* HEAD_NAMES[3] = {"points_head", "normal_head", "mask_head"};
static const int HEAD_BITS[3] = {MOGE_HEAD_POINTS, MOGE_HEAD_NORMAL, MOGE_HEAD_MASK};
static const int HEAD_COUT[3] = {3, 3, 1};
static const int KPATCH = 588, KPATCH_PAD = 640;     // 3*14*14 padded to a multiple of 64: the patch-embed GEMM then runs on the LDS-DMA kernels (K % 64 == 0) instead of the register-staged gemm_kernel (0.46 -> ~0.2 ms per 32 images)

struct TInfo { size_t off; int64_t numel; bool loaded; };

struct ProfRec { int cls; hipEvent_t e0, e1; double flops, bytes; };

struct moge_handle {
    moge_config cfg;            // MoGe-2 config; for a MoGe-1 handle only the ViT fields (embed_dim, depth, num_heads, n_taps, taps) and dims[0] (= dim_proj) are used
    float mask_thr = 0.5f;      // validity threshold on the mask output (v2: sigmoid probability > 0.5, v2.py:249; v1: raw > mask_threshold, v1.py:358)
    int version = 2;            // 1: moge.model.v1.MoGeModel (cfg1), 2: moge.model.v2.MoGeModel
    moge_v1_config cfg1;
    std::string bb = "encoder.backbone.";                        // state-dict prefix of the ViT
    const char* proj_fmt = "encoder.output_projections.%d";      // 1x1 projections of the taps (v1: "head.projects.%d")
    std::string mean_key = "encoder.image_mean", std_key = "encoder.image_std";
    int device;
    // fp32 master copy of the checkpoint (layout = f(config))
    std::map<std::string, TInfo> table;
    size_t master_floats = 0;
    float* master = nullptr;
    bool master_ready = false;
    long long* bcast_rec = nullptr;                              // 5-word status record of moge_broadcast_weights (allocated once, at create)
    // derived fp32 vectors (bias sums, uv columns, expanded convT biases)
    std::map<std::string, size_t> aux_off;
    size_t aux_floats = 0;
    float* aux = nullptr;
    bool aux_ready = false;
    // kernel-layout matrices per precision
    std::map<std::string, size_t> pk_off;      // element offsets (same for both precisions)
    size_t pk_elems = 0;
    void* packed[2] = {nullptr, nullptr};
    bool pk_ready[2] = {false, false};
    int prec = MOGE_FP32;       // MOGE_FP32 / MOGE_FP16: storage type of the kernels (index into packed[])
    bool half_resid = false;    // MOGE_FP16_HALF: prec == MOGE_FP16 with the residual stream in fp16 (`.half()` semantics)
    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    // position-embedding cache
    struct PosEntry { int rows, cols, mode; float* ptr; };
    int onnx_mode = 0;          // onnx_compatible_mode (v2.py:67-74): plain bilinear 14x resize, size-based pos-embed resampling
    std::vector<PosEntry> pos_cache;
    int* d_status = nullptr;
    int* h_status = nullptr;       // pinned host copy of d_status: moge_sync reads it back on the stream, in front of its one host wait
    // staging for uint8 HWC input (img_dtype 2): converted to the model dtype, CHW, before the forward
    void* u8_stage = nullptr;
    size_t u8_stage_bytes = 0;
    // batch-split execution: two internal streams run the two halves of a batch concurrently (tails of one half's kernels
    // and its HBM-bound kernels overlap the other half's MFMA kernels); joined on the caller's stream before post-processing
    static constexpr int MAX_SPLIT = 4;
    hipStream_t split_st[MAX_SPLIT] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_SPLIT] = {nullptr, nullptr, nullptr, nullptr};
    // head streams (small batches): after the neck the decoder heads are independent and their launches do not fill the chip; heads 1, 2
    // of (sub-)batch `slot` run on head_st[slot][0 / 1] with their own scratch buffers, head 0 stays on the (sub-)batch's stream
    hipStream_t head_st[MAX_SPLIT][3] = {};
    hipEvent_t ev_neck[MAX_SPLIT] = {}, ev_head[MAX_SPLIT][3] = {};
    // pipelined heads (round 6, HEAD_PIPE): EVERY head on its own stream, started level by level behind the neck level it reads (ev_lvl[slot][l] is recorded on the
    // (sub-)batch's stream when neck level l is complete) - at one image the neck's and the heads' launches are half-chip grids that run side by side
    hipEvent_t ev_lvl[MAX_SPLIT][MOGE_LEVELS] = {};
    float img_mean[3] = {0.485f, 0.456f, 0.406f}, img_std[3] = {0.229f, 0.224f, 0.225f};   // refreshed from the checkpoint buffers
    // profiler
    bool prof_on = false;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    moge_profile prof_acc;
    // last forward (debug taps)
    struct { bool valid = false; int prec, B, rows, cols; std::map<std::string, std::pair<size_t, std::pair<int64_t, int>>> bufs; } last;   // name -> (ws offset, (numel, kind 0=f32 1=T))
};

// ------------------------------------------------------------------------------------------------------------
// weight table
// ------------------------------------------------------------------------------------------------------------
static void tadd(moge_handle* h, const std::string& name, int64_t numel) {
    TInfo t{h->master_floats, numel, false};
    h->table[name] = t;
    h->master_floats += (size_t)((numel + 63) / 64 * 64);
}
static void aadd(moge_handle* h, const std::string& name, int64_t numel) {
    h->aux_off[name] = h->aux_floats;
    h->aux_floats += (size_t)((numel + 63) / 64 * 64);
}
static void padd(moge_handle* h, const std::string& name, int64_t numel) {
    h->pk_off[name] = h->pk_elems;
    h->pk_elems += (size_t)((numel + 63) / 64 * 64);
}
static std::string S(const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return buf;
}

static void build_tables_v2_decoder(moge_handle* h);
static void build_tables_v1_decoder(moge_handle* h);
static void build_tables(moge_handle* h) {
    const moge_config& c = h->cfg;
    const int D = c.embed_dim, c0 = c.dims[0];
    const std::string bb = h->bb;
    tadd(h, bb + "cls_token", D);
    tadd(h, bb + "pos_embed", (int64_t)(1 + 37 * 37) * D);
    tadd(h, bb + "patch_embed.proj.weight", (int64_t)D * KPATCH);
    tadd(h, bb + "patch_embed.proj.bias", D);
    padd(h, "patch.w", (int64_t)D * KPATCH_PAD);
    for (int i = 0; i < c.depth; i++) {
        const std::string p = bb + S("blocks.%d.", i);
        tadd(h, p + "norm1.weight", D); tadd(h, p + "norm1.bias", D);
        tadd(h, p + "attn.qkv.weight", (int64_t)3 * D * D); tadd(h, p + "attn.qkv.bias", 3 * D);
        tadd(h, p + "attn.proj.weight", (int64_t)D * D); tadd(h, p + "attn.proj.bias", D);
        tadd(h, p + "ls1.gamma", D);
        tadd(h, p + "norm2.weight", D); tadd(h, p + "norm2.bias", D);
        tadd(h, p + "mlp.fc1.weight", (int64_t)4 * D * D); tadd(h, p + "mlp.fc1.bias", 4 * D);
        tadd(h, p + "mlp.fc2.weight", (int64_t)4 * D * D); tadd(h, p + "mlp.fc2.bias", D);
        tadd(h, p + "ls2.gamma", D);
        padd(h, S("blk%d.qkv", i), (int64_t)3 * D * D);
        padd(h, S("blk%d.proj", i), (int64_t)D * D);
        padd(h, S("blk%d.fc1", i), (int64_t)4 * D * D);
        padd(h, S("blk%d.fc2", i), (int64_t)4 * D * D);
        // LN fold (fp16 path): norm1 folded into qkv, norm2 into fc1:  W' = g (.) W,  c = row sums of W',  b' = b + W beta
        padd(h, S("blk%d.qkvf", i), (int64_t)3 * D * D);
        padd(h, S("blk%d.fc1f", i), (int64_t)4 * D * D);
        aadd(h, S("blk%d.qkv.c", i), 3 * D); aadd(h, S("blk%d.qkv.bf", i), 3 * D);
        aadd(h, S("blk%d.fc1.c", i), 4 * D); aadd(h, S("blk%d.fc1.bf", i), 4 * D);
    }
    tadd(h, bb + "norm.weight", D); tadd(h, bb + "norm.bias", D);
    for (int k = 0; k < c.n_taps; k++) {
        tadd(h, S(h->proj_fmt, k) + ".weight", (int64_t)c0 * D);
        tadd(h, S(h->proj_fmt, k) + ".bias", c0);
    }
    tadd(h, h->mean_key, 3);
    tadd(h, h->std_key, 3);
    padd(h, "outproj.w", (int64_t)c0 * c.n_taps * D);
    aadd(h, "outproj.bias", c0);
    if (h->version == 1) build_tables_v1_decoder(h);
    else build_tables_v2_decoder(h);
}

// ConvStack options (ABI v3).  The released layout is what the fused throughput paths (side inputs, level-4 dot, composed chains) are built
// for; every other combination runs the same kernels un-fused.
static const int* stack_rs(const moge_config& c, bool neck) { return neck ? c.neck_resamplers : c.head_resamplers; }
static bool rs_is_phase(int r) { return r == MOGE_RS_BILINEAR || r == MOGE_RS_NEAREST; }
static const char* rs_final_conv(int r) { return r == MOGE_RS_PIXEL_SHUFFLE ? "2" : "1"; }      // index of the resampler's last 3x3 conv in its nn.Sequential
static bool stack_has_norm(const moge_config& c, bool neck) { return neck ? (c.neck_in_norm || c.neck_hidden_norm) : (c.head_in_norm || c.head_hidden_norm); }
static int stack_act(const moge_config& c, bool neck) { return neck ? c.neck_activation : c.head_activation; }
static int stack_mult(const moge_config& c, bool neck) { const int m = neck ? c.neck_hidden_mult : c.head_hidden_mult; return m > 0 ? m : 1; }      // dim_times_res_block_hidden (0 = unset)
// residual blocks that leave the fused [ReLU -> 3x3 -> ReLU -> 3x3] pair of launches: any norm, or an activation the conv kernels do not fuse
static bool stack_generic_blocks(const moge_config& c, bool neck) { return stack_has_norm(c, neck) || stack_act(c, neck) != MOGE_ACT_RELU; }
// groups of a residual-block norm on a C-channel map (modules.py:47-58)
static int norm_groups(int mode, int C) { return mode == MOGE_NORM_LAYER ? 1 : mode == MOGE_NORM_INSTANCE ? C : (C / 32 > 0 ? C / 32 : 1); }


// ---- CT3: ConvTranspose2d(k2, s2) + 3x3 composed (conv_pp.hip, round 6) ----------------------------------------------------------------------------
// Worth it where the composed conv is cheaper than the pair: 16 Cin Cout MACs per output pixel against 2 Cin Cout + 9 Cout^2 -> Cin = 2 Cout (the released layout's
// 256 -> 128 and 128 -> 64 resamplers; 1024 -> 256 would cost 23 % MORE).  One phase per column block of the kernel: Cout = 128 or 64.
static bool ct3_shape(int ci, int co) { return ci == 2 * co && (co == 128 || co == 64); }
// Term tables of the composed weight blocks (Ct3Slot, elementwise.hip ct3_combine_kernel).  P(k = ky * 3 + kx, s = sy * 2 + sx) = W3[:, :, ky, kx] . WT[:, :, sy, sx]^T.
// Interior, phase (py, px), tap (tdy, tdx) = low-res pixel (y + py - 1 + tdy, x + px - 1 + tdx): the 3x3 taps whose high-res row 2y + py + ky - 1 lies in that
// low-res row, with the parity it has there:   py = 0: tdy = 0 <- (ky 0, sy 1);  tdy = 1 <- (ky 1, sy 0), (ky 2, sy 1);   py = 1: tdy = 0 <- (ky 0, sy 0), (ky 1, sy 1);
// tdy = 1 <- (ky 2, sy 0) - and the same along x.
static void ct3_interior_slots(std::vector<Ct3Slot>& v) {
    auto members = [](int p, int td, int (&kk)[2], int (&ss)[2]) -> int {
        if (p == 0 && td == 0) { kk[0] = 0; ss[0] = 1; return 1; }
        if (p == 0 && td == 1) { kk[0] = 1; ss[0] = 0; kk[1] = 2; ss[1] = 1; return 2; }
        if (p == 1 && td == 0) { kk[0] = 0; ss[0] = 0; kk[1] = 1; ss[1] = 1; return 2; }
        kk[0] = 2; ss[0] = 0; return 1;
    };
    for (int py = 0; py < 2; py++) for (int px = 0; px < 2; px++) for (int tdy = 0; tdy < 2; tdy++) for (int tdx = 0; tdx < 2; tdx++) {
        Ct3Slot sl; memset(&sl, 0, sizeof(sl));
        sl.rowblk = py * 2 + px; sl.colblk = tdy * 2 + tdx;
        int ky[2], sy[2], kx[2], sx[2];
        const int ny = members(py, tdy, ky, sy), nx = members(px, tdx, kx, sx);
        for (int a = 0; a < ny; a++) for (int b2 = 0; b2 < nx; b2++) sl.term[sl.nterms++] = (ky[a] * 3 + kx[b2]) | ((sy[a] * 2 + sx[b2]) << 4);
        v.push_back(sl);
    }
}
// Border classes (conv_pp.hip ct3_border_kernel): exact (replicate-padded high-res map) minus composed (= reflecting pad), evaluated symbolically on a 4 x 4
// low-res map for one representative pixel per class; every term lands on one of the class's two cells.
static int ct3_border_slots(std::vector<Ct3Slot>& v) {
    const int H = 4, W = 4;
    struct Rep { int Y, X, cy0, cx0, cy1, cx1; };
    const Rep reps[12] = {{0, 4, 0, 1, 0, 2}, {0, 3, 0, 1, 0, 2}, {7, 4, 3, 1, 3, 2}, {7, 3, 3, 1, 3, 2},
                          {4, 0, 1, 0, 2, 0}, {3, 0, 1, 0, 2, 0}, {4, 7, 1, 3, 2, 3}, {3, 7, 1, 3, 2, 3},
                          {0, 0, 0, 0, -1, -1}, {0, 7, 0, 3, -1, -1}, {7, 0, 3, 0, -1, -1}, {7, 7, 3, 3, -1, -1}};
    for (int cls = 0; cls < 12; cls++) {
        const Rep& r = reps[cls];
        Ct3Slot sl[2]; memset(sl, 0, sizeof(sl));
        sl[0].rowblk = sl[1].rowblk = cls; sl[0].colblk = 0; sl[1].colblk = 1;
        for (int ky = 0; ky < 3; ky++) for (int kx = 0; kx < 3; kx++) {
            const int rr = r.Y + ky - 1, cc = r.X + kx - 1;
            const int r_rep = rr < 0 ? 0 : (rr > 2 * H - 1 ? 2 * H - 1 : rr), c_rep = cc < 0 ? 0 : (cc > 2 * W - 1 ? 2 * W - 1 : cc);
            const int r_ref = rr < 0 ? 1 : (rr > 2 * H - 1 ? 2 * H - 2 : rr), c_ref = cc < 0 ? 1 : (cc > 2 * W - 1 ? 2 * W - 2 : cc);
            if (r_rep == r_ref && c_rep == c_ref) continue;
            for (int side = 0; side < 2; side++) {            // + replicated, - reflected
                const int ry = side ? r_ref : r_rep, cx = side ? c_ref : c_rep;
                const int cy = ry >> 1, cxx = cx >> 1, sidx = (ry & 1) * 2 + (cx & 1);
                const int slot = (cy == r.cy0 && cxx == r.cx0) ? 0 : ((cy == r.cy1 && cxx == r.cx1) ? 1 : -1);
                if (slot < 0 || sl[slot].nterms >= 13) return -1;
                sl[slot].term[sl[slot].nterms++] = (ky * 3 + kx) | (sidx << 4) | (side ? 256 : 0);
            }
        }
        v.push_back(sl[0]); v.push_back(sl[1]);
    }
    return 0;
}

static void build_tables_v2_decoder(moge_handle* h) {
    const moge_config& c = h->cfg;
    const int D = c.embed_dim, c0 = c.dims[0];
    auto stack = [&](const std::string& name, bool neck, const int* nres, int cout) {
        const int* rs = stack_rs(c, neck);
        const int in_norm = neck ? c.neck_in_norm : c.head_in_norm, hid_norm = neck ? c.neck_hidden_norm : c.head_hidden_norm;
        for (int l = 0; l < MOGE_LEVELS; l++) {
            const int cl = c.dims[l];
            const int cin = neck ? (l == 0 ? c0 + 2 : 2) : cl;
            tadd(h, name + S(".input_blocks.%d.weight", l), (int64_t)cl * cin);
            tadd(h, name + S(".input_blocks.%d.bias", l), cl);
            if (neck) {
                aadd(h, name + S(".in%d.wu", l), cl);
                aadd(h, name + S(".in%d.wv", l), cl);
                if (l == 0) {
                    padd(h, name + ".in0.w", (int64_t)cl * c0);
                    // fp16 path: the summed output projections and this 1x1 block have no non-linearity between them (modules.py:128-131,
                    // 245): composed at pack time into ONE [c0][n_taps * D] matrix applied to the K-concatenated taps (COMPOSE, below)
                    padd(h, name + ".in0c.w", (int64_t)cl * c.n_taps * D);
                    aadd(h, name + ".in0c.bias", cl);
                }
                else aadd(h, name + S(".rs%d.bias2", l - 1), cl);
            } else {
                padd(h, name + S(".in%d.w", l), (int64_t)cl * cl);
            }
        }
        for (int l = 0; l < MOGE_LEVELS - 1; l++) {
            const int ci = c.dims[l], co = c.dims[l + 1];
            if (rs[l] == MOGE_RS_PIXEL_SHUFFLE) {
                // Conv2d(ci, 4 co, 3x3) -> PixelShuffle(2) -> Conv2d(co, co, 3x3)   (modules.py:146-151)
                tadd(h, name + S(".resamplers.%d.0.weight", l), (int64_t)4 * co * ci * 9);
                tadd(h, name + S(".resamplers.%d.0.bias", l), 4 * co);
                tadd(h, name + S(".resamplers.%d.2.weight", l), (int64_t)co * co * 9);
                tadd(h, name + S(".resamplers.%d.2.bias", l), co);
                padd(h, name + S(".rs%d.w3p", l), (int64_t)4 * co * 9 * ci);       // rows permuted to (dy, dx, co): the phase-conv layout
                padd(h, name + S(".rs%d.w3", l), (int64_t)co * 9 * co);
                aadd(h, name + S(".rs%d.bias4", l), 4 * co);
            } else if (rs[l] == MOGE_RS_CONV_TRANSPOSE) {
                tadd(h, name + S(".resamplers.%d.0.weight", l), (int64_t)ci * co * 4);
                tadd(h, name + S(".resamplers.%d.0.bias", l), co);
                tadd(h, name + S(".resamplers.%d.1.weight", l), (int64_t)co * co * 9);
                tadd(h, name + S(".resamplers.%d.1.bias", l), co);
                padd(h, name + S(".rs%d.wT", l), (int64_t)4 * co * ci);
                padd(h, name + S(".rs%d.w3", l), (int64_t)co * 9 * co);
                aadd(h, name + S(".rs%d.biasT", l), 4 * co);
                if (!neck) aadd(h, name + S(".rs%d.bias_in", l), co);     // resampler conv bias + next level's input-block bias (fused path)
                if (ct3_shape(ci, co)) {
                    // fp16 path (round 6): ConvTranspose2d + 3x3 composed into ONE 4-phase 2x2-tap conv on the low-res map (conv_pp.hip CT3) + its border weights
                    padd(h, name + S(".rs%d.wc", l), (int64_t)4 * co * 4 * ci);
                    padd(h, name + S(".rs%d.dw", l), (int64_t)12 * co * 2 * ci);
                    aadd(h, name + S(".rs%d.bias_ct3", l), 4 * co);
                }
                if (!neck && l == 0 && nres[0] == 0) {
                    // fp16 path: a head without level-0 residual blocks applies its ConvTranspose2d directly to its level-0 input block
                    // (modules.py:245-250): both linear, composed at pack time into one [4 co][c0] matrix on the neck's level-0 map
                    padd(h, name + ".rs0.wTc", (int64_t)4 * co * ci);
                    aadd(h, name + ".rs0.biasTc", 4 * co);
                }
            } else {
                tadd(h, name + S(".resamplers.%d.1.weight", l), (int64_t)co * ci * 9);
                tadd(h, name + S(".resamplers.%d.1.bias", l), co);
                padd(h, name + S(".rs%d.w3p", l), (int64_t)4 * co * 9 * ci);
                aadd(h, name + S(".rs%d.bias4", l), 4 * co);
            }
        }
        for (int l = 0; l < MOGE_LEVELS; l++)
            for (int j = 0; j < nres[l]; j++) {
                const int cl = c.dims[l], ch = cl * stack_mult(c, neck);          // hidden width (modules.py:222)
                // (InstanceNorm2d has no parameters: affine=False, track_running_stats=False - nothing in the state dict)
                if (in_norm && in_norm != MOGE_NORM_INSTANCE) { tadd(h, name + S(".res_blocks.%d.%d.layers.0.weight", l, j), cl); tadd(h, name + S(".res_blocks.%d.%d.layers.0.bias", l, j), cl); }
                if (hid_norm && hid_norm != MOGE_NORM_INSTANCE) { tadd(h, name + S(".res_blocks.%d.%d.layers.3.weight", l, j), ch); tadd(h, name + S(".res_blocks.%d.%d.layers.3.bias", l, j), ch); }
                for (int li = 2; li <= 5; li += 3) {
                    tadd(h, name + S(".res_blocks.%d.%d.layers.%d.weight", l, j, li), (int64_t)cl * ch * 9);
                    tadd(h, name + S(".res_blocks.%d.%d.layers.%d.bias", l, j, li), li == 2 ? ch : cl);
                    padd(h, name + S(".res%d.%d.w%d", l, j, li == 2 ? 1 : 2), (int64_t)cl * 9 * ch);
                }
            }
        if (cout > 0) {
            tadd(h, name + ".output_blocks.4.weight", (int64_t)cout * c.dims[4]);
            tadd(h, name + ".output_blocks.4.bias", cout);
            aadd(h, name + ".out4.w2", (int64_t)cout * c.dims[4]);      // output conv o level-4 input block (fp16 path, see head_final_kernel)
            aadd(h, name + ".out4.b2", cout);
            aadd(h, name + ".dot.own", 256);                            // conv_pp fused output conv: one group of A-operand rows (512 halves), Wout
        }
        if (neck) {
            aadd(h, "neck.dot.w2cat", 3 * 4 * 32);                      // the heads' (Wout . Win4) rows, four per head (zero padded), head order
            aadd(h, "neck.dot", 3 * 256);                               // ... as up to three A-operand groups
        }
    };
    stack("neck", true, c.neck_res_blocks, 0);
    for (int k = 0; k < 3; k++)
        if (c.heads & HEAD_BITS[k]) stack(HEAD_NAMES[k], false, c.head_res_blocks, HEAD_COUT[k]);
    if (c.heads & MOGE_HEAD_SCALE) {
        const int hd = c.scale_hidden;
        tadd(h, "scale_head.0.weight", (int64_t)hd * D); tadd(h, "scale_head.0.bias", hd);
        tadd(h, "scale_head.2.weight", (int64_t)hd * hd); tadd(h, "scale_head.2.bias", hd);
        tadd(h, "scale_head.4.weight", hd); tadd(h, "scale_head.4.bias", 1);
    }
}

static const float* M(moge_handle* h, const std::string& name) { return h->master + h->table.at(name).off; }
static float* A(moge_handle* h, const std::string& name) { return h->aux + h->aux_off.at(name); }
template <typename T> static const T* P(moge_handle* h, const std::string& name) { return reinterpret_cast<const T*>(h->packed[TT<T>::PREC]) + h->pk_off.at(name); }
template <typename T> static T* Pm(moge_handle* h, const std::string& name) { return reinterpret_cast<T*>(h->packed[TT<T>::PREC]) + h->pk_off.at(name); }

// ------------------------------------------------------------------------------------------------------------
// packing (device side, from the fp32 master)
// ------------------------------------------------------------------------------------------------------------
static int build_aux_v1(moge_handle* h, hipStream_t st);
static int build_aux(moge_handle* h, hipStream_t st) {
    if (h->aux_ready) return 0;
    const moge_config& c = h->cfg;
    if (!h->aux) HIPCHK(hipMalloc(&h->aux, h->aux_floats * sizeof(float)));
    HIPCHK(hipMemsetAsync(h->aux, 0, h->aux_floats * sizeof(float), st));
    const int c0 = c.dims[0];
    // outproj.bias = sum_k bias_k : n_taps strided adds via repack (dst += not available) -> do on host-visible path: tiny D2H/H2D
    {
        std::vector<float> acc(c0, 0.f), tmp(c0);
        for (int k = 0; k < c.n_taps; k++) {
            HIPCHK(hipMemcpyAsync(tmp.data(), M(h, S(h->proj_fmt, k) + ".bias"), c0 * sizeof(float), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (int i = 0; i < c0; i++) acc[i] += tmp[i];
        }
        HIPCHK(hipMemcpyAsync(A(h, "outproj.bias"), acc.data(), c0 * sizeof(float), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    if (h->version == 1) { CHK(build_aux_v1(h, st)); h->aux_ready = true; return 0; }
    // neck uv columns + combined biases
    for (int l = 0; l < MOGE_LEVELS; l++) {
        const int cl = c.dims[l];
        const int cin = l == 0 ? c0 + 2 : 2;
        const float* w = M(h, S("neck.input_blocks.%d.weight", l));
        LCHK(launch_repack<float>(w + (cin - 2), A(h, S("neck.in%d.wu", l)), cl, 1, 1, 1, cin, 0, 0, 0, 1, 0, 0, st));
        LCHK(launch_repack<float>(w + (cin - 1), A(h, S("neck.in%d.wv", l)), cl, 1, 1, 1, cin, 0, 0, 0, 1, 0, 0, st));
        if (l > 0) {
            std::vector<float> a(cl), b(cl);
            HIPCHK(hipMemcpyAsync(a.data(), M(h, S("neck.resamplers.%d.%s.bias", l - 1, rs_final_conv(c.neck_resamplers[l - 1]))), cl * sizeof(float), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(b.data(), M(h, S("neck.input_blocks.%d.bias", l)), cl * sizeof(float), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (int i = 0; i < cl; i++) a[i] += b[i];
            HIPCHK(hipMemcpyAsync(A(h, S("neck.rs%d.bias2", l - 1)), a.data(), cl * sizeof(float), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    auto stack = [&](const std::string& name) -> int {
        const bool neck = name == "neck";
        const int* rs = stack_rs(c, neck);
        for (int l = 0; l < MOGE_LEVELS - 1; l++) {
            const int co = c.dims[l + 1];
            if (rs[l] == MOGE_RS_CONV_TRANSPOSE) {
                LCHK(launch_repack<float>(M(h, name + S(".resamplers.%d.0.bias", l)), A(h, name + S(".rs%d.biasT", l)), 4, 1, 1, co, 0, 0, 0, 1, co, 0, 0, st));
                if (!neck) {
                    std::vector<float> a(co), b(co);
                    HIPCHK(hipMemcpyAsync(a.data(), M(h, name + S(".resamplers.%d.1.bias", l)), co * sizeof(float), hipMemcpyDeviceToHost, st));
                    HIPCHK(hipMemcpyAsync(b.data(), M(h, name + S(".input_blocks.%d.bias", l + 1)), co * sizeof(float), hipMemcpyDeviceToHost, st));
                    HIPCHK(hipStreamSynchronize(st));
                    for (int i = 0; i < co; i++) a[i] += b[i];
                    HIPCHK(hipMemcpyAsync(A(h, name + S(".rs%d.bias_in", l)), a.data(), co * sizeof(float), hipMemcpyHostToDevice, st));
                    HIPCHK(hipStreamSynchronize(st));
                }
                if (ct3_shape(c.dims[l], co)) {
                    // CT3: the ConvTranspose2d bias passes through all nine taps of the 3x3 (replicate padding: at the border too):
                    // bias_ct3 = (3x3 bias + next level's input-block bias) + sum_{m, k} W3[o][m][k] bT[m], once per phase
                    std::vector<float> w3((size_t)co * co * 9), bt(co), base(co), b4((size_t)4 * co);
                    HIPCHK(hipMemcpyAsync(w3.data(), M(h, name + S(".resamplers.%d.1.weight", l)), w3.size() * 4, hipMemcpyDeviceToHost, st));
                    HIPCHK(hipMemcpyAsync(bt.data(), M(h, name + S(".resamplers.%d.0.bias", l)), (size_t)co * 4, hipMemcpyDeviceToHost, st));
                    HIPCHK(hipMemcpyAsync(base.data(), neck ? A(h, S("neck.rs%d.bias2", l)) : A(h, name + S(".rs%d.bias_in", l)), (size_t)co * 4, hipMemcpyDeviceToHost, st));
                    HIPCHK(hipStreamSynchronize(st));
                    for (int o = 0; o < co; o++) {
                        double a = base[o];
                        for (int m = 0; m < co; m++) {
                            double t = 0;
                            for (int k9 = 0; k9 < 9; k9++) t += w3[((size_t)o * co + m) * 9 + k9];
                            a += t * bt[m];
                        }
                        for (int q = 0; q < 4; q++) b4[(size_t)q * co + o] = (float)a;
                    }
                    HIPCHK(hipMemcpyAsync(A(h, name + S(".rs%d.bias_ct3", l)), b4.data(), b4.size() * 4, hipMemcpyHostToDevice, st));
                    HIPCHK(hipStreamSynchronize(st));
                }
            } else if (rs_is_phase(rs[l])) {
                // up-sample x2 + 3x3 as a 4-phase conv: bias replicated per phase; the neck adds its next level's input-block bias (rs%d.bias2)
                const float* b4 = neck ? A(h, S("neck.rs%d.bias2", l)) : M(h, name + S(".resamplers.%d.1.bias", l));
                LCHK(launch_repack<float>(b4, A(h, name + S(".rs%d.bias4", l)), 4, 1, 1, co, 0, 0, 0, 1, co, 0, 0, st));
            } else {
                // pixel_shuffle: bias of the first conv, permuted from PixelShuffle's (co, dy, dx) channel order to (dy, dx, co)
                LCHK(launch_repack<float>(M(h, name + S(".resamplers.%d.0.bias", l)), A(h, name + S(".rs%d.bias4", l)), 4, 1, 1, co, 1, 0, 0, 4, co, 0, 0, st));
            }
        }
        return 0;
    };
    CHK(stack("neck"));
    int ndot = 0;                                   // heads that have a group in the neck's fused-output-conv table
    for (int k = 0; k < 3; k++)
        if (c.heads & HEAD_BITS[k]) {
            CHK(stack(HEAD_NAMES[k]));
            // W2 = Wout . Win,  b2 = bout + Wout . bin   (level 4: out(x + in_4(n4)) = Wout x + W2 n4 + b2)
            const std::string name = HEAD_NAMES[k];
            const int c4 = c.dims[4], co = HEAD_COUT[k];
            std::vector<float> wout((size_t)co * c4), win((size_t)c4 * c4), bin(c4), bout(co), w2((size_t)co * c4), b2(co);
            HIPCHK(hipMemcpyAsync(wout.data(), M(h, name + ".output_blocks.4.weight"), wout.size() * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(bout.data(), M(h, name + ".output_blocks.4.bias"), bout.size() * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(win.data(), M(h, name + ".input_blocks.4.weight"), win.size() * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(bin.data(), M(h, name + ".input_blocks.4.bias"), bin.size() * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (int o = 0; o < co; o++) {
                double bb = bout[o];
                for (int m = 0; m < c4; m++) bb += (double)wout[(size_t)o * c4 + m] * bin[m];
                b2[o] = (float)bb;
                for (int j = 0; j < c4; j++) {
                    double a = 0.0;
                    for (int m = 0; m < c4; m++) a += (double)wout[(size_t)o * c4 + m] * win[(size_t)m * c4 + j];
                    w2[(size_t)o * c4 + j] = (float)a;
                }
            }
            HIPCHK(hipMemcpyAsync(A(h, name + ".out4.w2"), w2.data(), w2.size() * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(A(h, name + ".out4.b2"), b2.data(), b2.size() * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
            // fused output conv of level 4 (conv_pp.hip DOT, fp16 path): this head's Wout as its own group, its W2 as group `ndot` of the neck's table
            if (c4 == 32 && co <= 4) {
                LCHK(launch_pack_dot_table(M(h, name + ".output_blocks.4.weight"), co, 1, A(h, name + ".dot.own"), st));
                std::vector<float> rows4(4 * 32, 0.f);
                for (int o = 0; o < co; o++)
                    for (int j = 0; j < 32; j++) rows4[o * 32 + j] = w2[(size_t)o * c4 + j];
                HIPCHK(hipMemcpyAsync(A(h, "neck.dot.w2cat") + ndot * 128, rows4.data(), 128 * 4, hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
                ndot++;
            }
        }
    if (ndot > 0) LCHK(launch_pack_dot_table(A(h, "neck.dot.w2cat"), 4 * ndot, ndot, A(h, "neck.dot"), st));
    // ---- composed linear chains (fp16 path, COMPOSE): the bias vectors, in double on the host (one-off, a few MB of D2H) ----
    {   // neck level 0: in0(sum_k proj_k(tap_k)) = (Win0 Wout) tapcat + (Win0 sum_k b_k + b_in0) + uv term
        std::vector<float> win((size_t)c0 * (c0 + 2)), bo(c0), bi(c0), bc(c0);
        HIPCHK(hipMemcpyAsync(win.data(), M(h, "neck.input_blocks.0.weight"), win.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(bo.data(), A(h, "outproj.bias"), (size_t)c0 * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(bi.data(), M(h, "neck.input_blocks.0.bias"), (size_t)c0 * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (int o = 0; o < c0; o++) {
            double a = bi[o];
            for (int j = 0; j < c0; j++) a += (double)win[(size_t)o * (c0 + 2) + j] * bo[j];
            bc[o] = (float)a;
        }
        HIPCHK(hipMemcpyAsync(A(h, "neck.in0c.bias"), bc.data(), (size_t)c0 * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    if (c.head_res_blocks[0] == 0 && c.head_resamplers[0] == MOGE_RS_CONV_TRANSPOSE)
        for (int k = 0; k < 3; k++)
            if (c.heads & HEAD_BITS[k]) {
                // head level 0 -> 1: convT(in0(n0)) = (WT Win0) n0 + (WT b_in0 + bT); torch ConvTranspose2d weight [ci][co][2][2]
                const std::string name = HEAD_NAMES[k];
                const int ci = c0, co = c.dims[1];
                std::vector<float> wt((size_t)ci * co * 4), bin(ci), bt(co), bc((size_t)4 * co);
                HIPCHK(hipMemcpyAsync(wt.data(), M(h, name + ".resamplers.0.0.weight"), wt.size() * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpyAsync(bin.data(), M(h, name + ".input_blocks.0.bias"), (size_t)ci * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpyAsync(bt.data(), M(h, name + ".resamplers.0.0.bias"), (size_t)co * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                for (int q = 0; q < 4; q++)
                    for (int o = 0; o < co; o++) {
                        double a = bt[o];
                        for (int m = 0; m < ci; m++) a += (double)wt[((size_t)m * co + o) * 4 + q] * bin[m];
                        bc[(size_t)q * co + o] = (float)a;
                    }
                HIPCHK(hipMemcpyAsync(A(h, name + ".rs0.biasTc"), bc.data(), bc.size() * 4, hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
            }
    h->aux_ready = true;
    return 0;
}

// C[M][N] = A[M][K] * Bm[K][N], all fp32 row-major on the device, in exact fp32 (gemm.hip's v_mfma_f32_32x32x2_f32 path).  `scr` holds
// M * K + N * K floats (A made contiguous, Bm transposed: the GEMM kernels contract K-contiguous rows of both operands).
static int dev_matmul_f32(const float* Am, long lda, const float* Bm, long ldb, float* Cm, int Mr, int Nc, int Kc, float* scr, hipStream_t st) {
    float* Ac = scr;
    float* Bt = scr + (size_t)Mr * Kc;
    LCHK(launch_repack<float>(Am, Ac, Mr, 1, 1, Kc, lda, 0, 0, 1, Kc, 0, 0, st));
    LCHK(launch_repack<float>(Bm, Bt, Nc, 1, 1, Kc, 1, 0, 0, ldb, Kc, 0, 0, st));         // Bt[n][k] = Bm[k][n]
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.a = Ac; g.lda = Kc; g.w = Bt; g.ldw = Kc; g.M = Mr; g.N = Nc; g.K = Kc; g.epi = EPI_STORE; g.out = Cm; g.ldc = Nc;
    LCHK(launch_gemm<float>(g, AMODE_LINEAR, st));
    return 0;
}


// CT3 weights of one ConvTranspose2d + 3x3 resampler (fp16 path): P = the 36 basic products in ONE exact-fp32 GEMM ([9 Cout][Cout] x [Cout][4 Cin]), then the
// interior blocks -> wc [4 Cout][4 Cin] and the border blocks -> dw [12][Cout][2 Cin] as signed sums of them, one rounding to fp16 each.
// w3 = Conv2d weight [co][co][3][3], wt = ConvTranspose2d weight [ci][co][2][2] (fp32, device).  Also the test entry's (test_api.hip).
int ct3_compose_device(const float* w3, const float* wt, int ci, int co, void* wc, void* dw, hipStream_t st) {
    std::vector<Ct3Slot> slots;
    ct3_interior_slots(slots);
    const int n_int = (int)slots.size();
    if (ct3_border_slots(slots) != 0) return -2;
    const size_t nA = (size_t)9 * co * co, nB = (size_t)4 * ci * co, nP = (size_t)9 * co * 4 * ci;
    float* arena = nullptr;
    if (hipMalloc(&arena, (nA + nB + nP) * sizeof(float) + slots.size() * sizeof(Ct3Slot)) != hipSuccess) return -3;
    float* A9 = arena; float* Bt = A9 + nA; float* P = Bt + nB;
    Ct3Slot* dsl = reinterpret_cast<Ct3Slot*>(P + nP);
    int rc = 0;
    hipError_t e = hipMemcpyAsync(dsl, slots.data(), slots.size() * sizeof(Ct3Slot), hipMemcpyHostToDevice, st);
    do {
        if (e != hipSuccess) { rc = (int)e; break; }
        // A9[(k, o)][m] = W3[o][m][k];  Bt[(s, i)][m] = WT[i][m][s]   (both K-contiguous rows: what the GEMM kernels contract)
        if ((rc = launch_repack<float>(w3, A9, 9, co, 1, co, 1, (long)co * 9, 0, 9, (long)co * co, co, 0, st))) break;
        if ((rc = launch_repack<float>(wt, Bt, 4, ci, 1, co, 1, (long)co * 4, 0, 4, (long)ci * co, co, 0, st))) break;
        GemmArgs g; memset(&g, 0, sizeof(g));
        g.a = A9; g.lda = co; g.w = Bt; g.ldw = co; g.M = 9 * co; g.N = 4 * ci; g.K = co; g.epi = EPI_STORE; g.out = P; g.ldc = 4 * ci;
        if ((rc = launch_gemm<float>(g, AMODE_LINEAR, st))) break;
        if ((rc = launch_ct3_combine(P, dsl, n_int, co, ci, wc, 4 * ci, st))) break;
        if ((rc = launch_ct3_combine(P, dsl + n_int, (int)slots.size() - n_int, co, ci, dw, 2 * ci, st))) break;
    } while (0);
    hipError_t e2 = hipStreamSynchronize(st);      // (the host-side term table must outlive its copy; pack time, not the hot path)
    hipFree(arena);
    if (rc) return rc;
    return e2 == hipSuccess ? 0 : (int)e2;
}
static int compose_ct3_f16(moge_handle* h, const std::string& name, int l, int ci, int co, hipStream_t st) {
    const int rc = ct3_compose_device(M(h, name + S(".resamplers.%d.1.weight", l)), M(h, name + S(".resamplers.%d.0.weight", l)), ci, co,
                                      Pm<f16>(h, name + S(".rs%d.wc", l)), Pm<f16>(h, name + S(".rs%d.dw", l)), st);
    if (rc) return fail(rc < 0 ? MOGE_ERR_INVALID : MOGE_ERR_HIP, "composing the ConvTranspose2d + 3x3 weights of %s level %d failed (%d)", name.c_str(), l, rc);
    return 0;
}

// Composed linear chains of the fp16 path (COMPOSE): products formed in fp32 from the master weights, ONE rounding to fp16.
static int compose_weights_f16(moge_handle* h, hipStream_t st) {
    const moge_config& c = h->cfg;
    const int D = c.embed_dim, c0 = c.dims[0], K4 = c.n_taps * D, co = c.dims[1];
    const size_t nscr = (size_t)c0 * c0 + (size_t)K4 * c0 + (size_t)4 * co * c0 + (size_t)c0 * c0;
    const size_t ntmp = (size_t)c0 * K4 > (size_t)4 * co * c0 ? (size_t)c0 * K4 : (size_t)4 * co * c0;
    // one temporary arena (freed on every path below): scr | tmp | wcat | wT
    float* arena = nullptr;
    HIPCHK(hipMalloc(&arena, (nscr + ntmp + (size_t)c0 * K4 + (size_t)4 * co * c0) * 4));
    float* scr = arena;
    float* tmp = scr + nscr;
    float* wcat = tmp + ntmp;
    float* wT = wcat + (size_t)c0 * K4;
    int rc = 0;
    do {
        // Wout_cat [c0][n_taps * D] (fp32), then (Win0[:, :c0]) . Wout_cat
        for (int k = 0; k < c.n_taps && !rc; k++)
            rc = launch_repack<float>(M(h, S(h->proj_fmt, k) + ".weight"), wcat + (size_t)k * D, c0, 1, 1, D, D, 0, 0, 1, K4, 0, 0, st);
        if (rc) break;
        if ((rc = dev_matmul_f32(M(h, "neck.input_blocks.0.weight"), c0 + 2, wcat, K4, tmp, c0, K4, c0, scr, st))) break;
        if ((rc = launch_convert<float, f16>(tmp, Pm<f16>(h, "neck.in0c.w"), (long)c0 * K4, st))) break;
        if (c.head_res_blocks[0] == 0 && c.head_resamplers[0] == MOGE_RS_CONV_TRANSPOSE)
            for (int k = 0; k < 3 && !rc; k++)
                if (c.heads & HEAD_BITS[k]) {
                    const std::string name = HEAD_NAMES[k];
                    // WT [(dy*2+dx)*co + o][ci] (fp32) . Win0_head [ci][ci]
                    rc = launch_repack<float>(M(h, name + ".resamplers.0.0.weight"), wT, 4, co, 1, c0, 1, 4, 0, (long)co * 4, (long)co * c0, c0, 0, st);
                    if (!rc) rc = dev_matmul_f32(wT, c0, M(h, name + ".input_blocks.0.weight"), c0, tmp, 4 * co, c0, c0, scr, st);
                    if (!rc) rc = launch_convert<float, f16>(tmp, Pm<f16>(h, name + ".rs0.wTc"), (long)4 * co * c0, st);
                }
    } while (0);
    hipError_t e = hipStreamSynchronize(st);
    hipFree(arena);
    if (rc) return fail(rc < 0 ? MOGE_ERR_INVALID : MOGE_ERR_HIP, "composing the level-0 linear chains failed (%d)", rc);
    HIPCHK(e);
    return 0;
}

template <typename T> static int pack_weights_v1(moge_handle* h, hipStream_t st);
template <typename T>
static int pack_weights(moge_handle* h, hipStream_t st) {
    const int pr = TT<T>::PREC;
    if (h->pk_ready[pr]) return 0;
    const moge_config& c = h->cfg;
    const int D = c.embed_dim, c0 = c.dims[0];
    if (!h->packed[pr]) HIPCHK(hipMalloc(&h->packed[pr], h->pk_elems * sizeof(T)));
    HIPCHK(hipMemsetAsync(h->packed[pr], 0, h->pk_elems * sizeof(T), st));
    const std::string bb = h->bb;
    // patch embed [D][588] -> [D][640]
    LCHK(launch_repack<T>(M(h, bb + "patch_embed.proj.weight"), Pm<T>(h, "patch.w"), D, 1, 1, KPATCH, KPATCH, 0, 0, 1, KPATCH_PAD, 0, 0, st));
    for (int i = 0; i < c.depth; i++) {
        const std::string p = bb + S("blocks.%d.", i);
        LCHK((launch_convert<float, T>(M(h, p + "attn.qkv.weight"), Pm<T>(h, S("blk%d.qkv", i)), (long)3 * D * D, st)));
        LCHK((launch_convert<float, T>(M(h, p + "attn.proj.weight"), Pm<T>(h, S("blk%d.proj", i)), (long)D * D, st)));
        LCHK((launch_convert<float, T>(M(h, p + "mlp.fc1.weight"), Pm<T>(h, S("blk%d.fc1", i)), (long)4 * D * D, st)));
        LCHK((launch_convert<float, T>(M(h, p + "mlp.fc2.weight"), Pm<T>(h, S("blk%d.fc2", i)), (long)4 * D * D, st)));
        if constexpr (std::is_same<T, f16>::value) {
            LCHK(launch_fold_ln<f16>(M(h, p + "attn.qkv.weight"), M(h, p + "norm1.weight"), M(h, p + "norm1.bias"), M(h, p + "attn.qkv.bias"),
                                     Pm<T>(h, S("blk%d.qkvf", i)), A(h, S("blk%d.qkv.c", i)), A(h, S("blk%d.qkv.bf", i)), 3 * D, D, st));
            LCHK(launch_fold_ln<f16>(M(h, p + "mlp.fc1.weight"), M(h, p + "norm2.weight"), M(h, p + "norm2.bias"), M(h, p + "mlp.fc1.bias"),
                                     Pm<T>(h, S("blk%d.fc1f", i)), A(h, S("blk%d.fc1.c", i)), A(h, S("blk%d.fc1.bf", i)), 4 * D, D, st));
        }
    }
    // output projections: [c0][D] x n_taps -> [c0][n_taps*D]
    for (int k = 0; k < c.n_taps; k++)
        LCHK(launch_repack<T>(M(h, S(h->proj_fmt, k) + ".weight"), Pm<T>(h, "outproj.w") + (size_t)k * D, c0, 1, 1, D, D, 0, 0, 1,
                              (long)c.n_taps * D, 0, 0, st));
    if (h->version == 1) { CHK(pack_weights_v1<T>(h, st)); h->pk_ready[pr] = true; return 0; }
    // neck.in0: [c0][c0+2] -> [c0][c0]
    LCHK(launch_repack<T>(M(h, "neck.input_blocks.0.weight"), Pm<T>(h, "neck.in0.w"), c0, 1, 1, c0, c0 + 2, 0, 0, 1, c0, 0, 0, st));
    auto conv3 = [&](const float* w, T* dst, int co, int ci) -> int {
        // torch [co][ci][3][3] -> [co][tap*ci + ci]
        return launch_repack<T>(w, dst, co, 9, 1, ci, (long)ci * 9, 1, 0, 9, (long)9 * ci, ci, 0, st);
    };
    auto stack = [&](const std::string& name, bool neck, const int* nres) -> int {
        for (int l = 0; l < MOGE_LEVELS; l++)
            if (!neck) LCHK((launch_convert<float, T>(M(h, name + S(".input_blocks.%d.weight", l)), Pm<T>(h, name + S(".in%d.w", l)), (long)c.dims[l] * c.dims[l], st)));
        const int* rs = stack_rs(c, neck);
        for (int l = 0; l < MOGE_LEVELS - 1; l++) {
            const int ci = c.dims[l], co = c.dims[l + 1];
            if (rs[l] == MOGE_RS_CONV_TRANSPOSE) {
                // ConvTranspose2d weight [ci][co][2][2] -> [(dy*2+dx)*co + o][ci]
                LCHK(launch_repack<T>(M(h, name + S(".resamplers.%d.0.weight", l)), Pm<T>(h, name + S(".rs%d.wT", l)), 4, co, 1, ci, 1, 4, 0, (long)co * 4,
                                      (long)co * ci, ci, 0, st));
                LCHK(conv3(M(h, name + S(".resamplers.%d.1.weight", l)), Pm<T>(h, name + S(".rs%d.w3", l)), co, co));
            } else if (rs_is_phase(rs[l])) {
                LCHK(launch_pack_phase_conv<T>(M(h, name + S(".resamplers.%d.1.weight", l)), Pm<T>(h, name + S(".rs%d.w3p", l)), co, ci, st, rs[l] == MOGE_RS_NEAREST ? 1 : 0));
            } else {
                // pixel_shuffle: Conv2d weight [(o*4 + q)][ci][3][3] -> [(q*co + o)][tap*ci + c]   (q = dy*2 + dx of nn.PixelShuffle(2))
                LCHK(launch_repack<T>(M(h, name + S(".resamplers.%d.0.weight", l)), Pm<T>(h, name + S(".rs%d.w3p", l)), 4, co, 9, ci, (long)ci * 9, (long)4 * ci * 9, 1, 9,
                                      (long)co * 9 * ci, (long)9 * ci, ci, st));
                LCHK(conv3(M(h, name + S(".resamplers.%d.2.weight", l)), Pm<T>(h, name + S(".rs%d.w3", l)), co, co));
            }
        }
        for (int l = 0; l < MOGE_LEVELS; l++)
            for (int j = 0; j < nres[l]; j++) {
                const int ch = c.dims[l] * stack_mult(c, neck);
                LCHK(conv3(M(h, name + S(".res_blocks.%d.%d.layers.2.weight", l, j)), Pm<T>(h, name + S(".res%d.%d.w1", l, j)), ch, c.dims[l]));
                LCHK(conv3(M(h, name + S(".res_blocks.%d.%d.layers.5.weight", l, j)), Pm<T>(h, name + S(".res%d.%d.w2", l, j)), c.dims[l], ch));
            }
        return 0;
    };
    CHK(stack("neck", true, c.neck_res_blocks));
    for (int k = 0; k < 3; k++)
        if (c.heads & HEAD_BITS[k]) CHK(stack(HEAD_NAMES[k], false, c.head_res_blocks));
    if constexpr (std::is_same<T, f16>::value) {
        CHK(compose_weights_f16(h, st));
        for (int kk = -1; kk < 3; kk++) {                                 // neck, then the heads
            if (kk >= 0 && !(c.heads & HEAD_BITS[kk])) continue;
            const std::string name = kk < 0 ? "neck" : HEAD_NAMES[kk];
            const int* rs = stack_rs(c, kk < 0);
            for (int l = 0; l < MOGE_LEVELS - 1; l++)
                if (rs[l] == MOGE_RS_CONV_TRANSPOSE && ct3_shape(c.dims[l], c.dims[l + 1])) CHK(compose_ct3_f16(h, name, l, c.dims[l], c.dims[l + 1], st));
        }
    }
    h->pk_ready[pr] = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// workspace plan
// ------------------------------------------------------------------------------------------------------------
struct Plan {
    size_t total = 0;
    size_t base = 0;          // byte offset of this plan inside the workspace arena (batch-split mode places two plans side by side)
    size_t patches, x, xn, q, k, vT, attn, hidden, tapcat, cls, mlp1, mlp2, metric, feat, neck[MOGE_LEVELS], scratch[12];
    int head_sets = 1;        // scratch triples: one per decoder head when the heads run on their own streams (small batches), else 1
    int slot = 0;             // index of this (sub-)batch among the batch-split parts (selects the head streams)
    size_t ln_part, ln_mr;    // LN fold: (sum, sum of squares) per row and 32-column group; (mean, rstd) per row
    size_t ln_cnt = 0, ln_cnt_bytes = 0;        // fused LN finalize of the latency-regime GEMMs: one counter per 64 rows (zero before the first launch, left zero)
    size_t attn_ws = 0, attn_ws_bytes = 0;      // stream-K attention (sub-round grids, fp16): counters + segment slots (attention_pp.hip); 0 = plain grid
    size_t maskprob, focal, shift, intr, pts_tmp, nrm_tmp, post_end;
    size_t gn = 0;            // GroupNorm partial sums (normalised residual blocks, ABI v3)
    int B, H, W, rows, cols, Np, Ntok, Npad;
    size_t scratch_elems;
};
static size_t take(Plan& p, size_t bytes) {
    const size_t off = p.total;
    p.total += (bytes + 255) / 256 * 256;
    return off;
}
static Plan make_plan(const moge_config& c, int prec, int B, int H, int W, int rows, int cols) {
    Plan p;
    const size_t s = prec == MOGE_FP16 ? 2 : 4;
    const int D = c.embed_dim;
    p.B = B; p.H = H; p.W = W; p.rows = rows; p.cols = cols;
    p.Np = rows * cols; p.Ntok = p.Np + 1; p.Npad = (p.Ntok + 63) / 64 * 64;
    const size_t BN = (size_t)B * p.Ntok, BP = (size_t)B * p.Np;
    // caller-visible post-processing buffers first (batch-split mode keeps these from the full-batch plan)
    const size_t px = (size_t)B * H * W;
    p.maskprob = take(p, px * 4);
    p.pts_tmp = take(p, px * 12);
    p.nrm_tmp = take(p, px * 12);
    p.focal = take(p, (size_t)B * 4);
    p.shift = take(p, (size_t)B * 4);
    p.intr = take(p, (size_t)B * 36);
    p.metric = take(p, (size_t)B * 4);
    p.post_end = p.total;
    p.patches = take(p, BP * KPATCH_PAD * s);
    p.x = take(p, BN * D * 4);
    p.xn = take(p, BN * D * s);
    p.ln_part = take(p, BN * (size_t)(D / 32) * 8);
    p.ln_mr = take(p, BN * 8);
    p.q = take(p, BN * D * s);
    p.k = take(p, BN * D * s);
    p.vT = take(p, (size_t)B * D * p.Npad * s);
    p.attn = take(p, BN * D * s);
    p.hidden = take(p, BN * 4 * D * s);
    if (prec == MOGE_FP16) {      // (adjacent: one memset zeroes the LN counters and the attention counters at the head of attn_ws)
        p.ln_cnt_bytes = (BN / 64 + 2) * 4; p.ln_cnt = take(p, p.ln_cnt_bytes);
        p.attn_ws_bytes = attention_pp_ws_bytes(B, c.num_heads, p.Ntok); p.attn_ws = take(p, p.attn_ws_bytes);
    }
    p.tapcat = take(p, BP * c.n_taps * D * s);
    p.cls = take(p, (size_t)B * D * 4);
    p.mlp1 = take(p, (size_t)B * (c.scale_hidden > 0 ? c.scale_hidden : 1) * 4);
    p.mlp2 = take(p, (size_t)B * (c.scale_hidden > 0 ? c.scale_hidden : 1) * 4);
    p.feat = take(p, BP * c.dims[0] * s);
    size_t mx = 0;
    for (int l = 0; l < MOGE_LEVELS; l++) {
        const size_t e = BP * ((size_t)1 << (2 * l)) * c.dims[l];
        p.neck[l] = take(p, e * s);
        if (e > mx) mx = e;
    }
    {   // the hidden map of a residual block is dim_times_res_block_hidden x its level's width (modules.py:222)
        const int km = stack_mult(c, true) > stack_mult(c, false) ? stack_mult(c, true) : stack_mult(c, false);
        mx *= (size_t)km;
    }
    p.scratch_elems = mx;
    int nheads = 0;
    for (int k = 0; k < 3; k++) nheads += (c.heads & HEAD_BITS[k]) ? 1 : 0;
    // (normalised residual blocks share ONE GroupNorm scratch per plan: heads then run one after the other)
    p.head_sets = (nheads > 1 && moge_tune_get("HEAD_STREAMS", 1) != 0 && B <= moge_tune_get("HEAD_STREAMS_MAX_B", 1) && !stack_has_norm(c, false)) ? nheads : 1;      // (only the shared norm-statistics scratch `gn` serialises the heads: activation-only blocks keep their streams)
    // (head streams: one triple more than heads - in the pipelined form triple 0 stays with the neck, which is still running when the first head starts)
    for (int i = 0; i < 3 * (p.head_sets > 1 ? p.head_sets + 1 : 1); i++) p.scratch[i] = take(p, mx * s);
    if (stack_has_norm(c, true) || stack_has_norm(c, false)) {
        size_t gmax = 0;
        for (int l = 0; l < MOGE_LEVELS; l++)
            for (int nk = 0; nk < 2; nk++) {
                const bool neck = nk == 1;
                const int nin = neck ? c.neck_in_norm : c.head_in_norm, nhid = neck ? c.neck_hidden_norm : c.head_hidden_norm;
                const int G1 = nin ? norm_groups(nin, c.dims[l]) : 0, G2 = nhid ? norm_groups(nhid, c.dims[l] * stack_mult(c, neck)) : 0;
                const size_t g1 = groupnorm_scratch_floats(B, rows << l, cols << l, G1 > G2 ? G1 : G2);
                if (g1 > gmax) gmax = g1;
            }
        p.gn = take(p, gmax * 4);
    }
    return p;
}
static size_t forward_ws_bytes(moge_handle* h, const Plan& pl);
static int ensure_ws(moge_handle* h, size_t bytes) {
    if (bytes <= h->ws_bytes) return 0;
    if (h->ws) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(h->ws)); h->ws = nullptr; h->ws_bytes = 0; }
    HIPCHK(hipMalloc(&h->ws, bytes));
    h->ws_bytes = bytes;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// profiler
// ------------------------------------------------------------------------------------------------------------
static hipEvent_t ev_get(moge_handle* h) {
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
struct ProfScope {
    moge_handle* h; hipStream_t st; ProfRec r; bool on;
    ProfScope(moge_handle* h_, hipStream_t st_, int cls, double flops, double bytes) : h(h_), st(st_), on(h_->prof_on) {
        if (on) { r.cls = cls; r.flops = flops; r.bytes = bytes; r.e0 = ev_get(h); r.e1 = ev_get(h); hipEventRecord(r.e0, st); }
    }
    ~ProfScope() { if (on) { hipEventRecord(r.e1, st); h->prof_pending.push_back(r); } }
};

// ------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------
static GemmArgs gemm_args() { GemmArgs g; memset(&g, 0, sizeof(g)); return g; }

static UVTerm uv_term(const float* wu, const float* wv, int pixW, int pixH, double aspect) {
    UVTerm u;
    const double sx = aspect / std::sqrt(1 + aspect * aspect), sy = 1 / std::sqrt(1 + aspect * aspect);
    u.wu = wu; u.wv = wv;
    u.u0 = (float)(-sx * (pixW - 1) / pixW); u.u1 = (float)(sx * (pixW - 1) / pixW);
    u.v0 = (float)(-sy * (pixH - 1) / pixH); u.v1 = (float)(sy * (pixH - 1) / pixH);
    u.ustep = pixW > 1 ? (u.u1 - u.u0) / (float)(pixW - 1) : 0.f;
    u.vstep = pixH > 1 ? (u.v1 - u.v0) / (float)(pixH - 1) : 0.f;
    return u;
}

template <typename T>
static int run_gemm(moge_handle* h, const GemmArgs& g, int amode, int cls, hipStream_t st, double kalgo = 0) {
    const double k = kalgo > 0 ? kalgo : (double)g.K;
    const double flops = 2.0 * g.M * (double)g.N * k;
    // COMPULSORY bytes of the launch (what an ideal kernel moves once): operands + weights + every output / read-modify-write the epilogue owns.
    // A 3x3 conv reads its input MAP once (M x C, not the 9x im2col rows); the fused side input is a second map; a residual add reads one more.
    const double e = sizeof(T);
    double bytes = (double)g.N * k * e;                                                       // weights
    if (amode == AMODE_CONV3) bytes += (double)g.M * g.C * e * (g.a2 ? 2 : 1);                 // input map (+ side map)
    else bytes += (double)g.M * k * e;                                                        // A rows
    switch (g.epi) {
    case EPI_RESID: bytes += (double)g.M * g.N * (g.xres ? 8.0 + (g.x16 ? 2.0 : 0.0) : 4.0) + (g.ln_part ? (double)g.M * (g.N / 32) * 8.0 : 0.0); break;   // fp32 x read + write, fp16 copy (fp16 stream: read + write), LN partials
    case EPI_PATCH: bytes += (double)g.M * g.N * 8.0; break;                                  // + pos read, fp32 x write
    default:
        if (g.dot_tab) bytes += (double)g.M * 4 * (16.0 * g.dot_nd);                           // fused output conv: 4 phases x 4 nd floats per low-res pixel
        else bytes += (double)g.M * g.N * e * (g.add ? 2 : 1);                                 // STORE / QKV / CONVT: the outputs (+ the added map)
        break;
    }
    if (cls == MOGE_KC_GEMM && amode == AMODE_LINEAR && std::is_same<T, f16>::value && gemm_runs_pp(g)) cls = MOGE_KC_GEMM_PP;
    ProfScope ps(h, st, cls, flops, bytes);
    LCHK(launch_gemm<T>(g, amode, st));
    return 0;
}

template <typename T>
static int conv3x3(moge_handle* h, const T* in, const T* w, const float* bias, T* out, int B, int Hh, int Ww, int Cin, int Cout, int relu_in,
                   int act, const T* add, const UVTerm* uv, hipStream_t st, const T* side = nullptr, const T* side_w = nullptr, bool* fused = nullptr) {
    GemmArgs g = gemm_args();
    g.a = in; g.H = Hh; g.W = Ww; g.C = Cin; g.relu_in = relu_in;
    g.w = w; g.ldw = 9 * Cin;
    g.M = B * Hh * Ww; g.N = Cout; g.K = 9 * Cin;
    g.epi = EPI_STORE; g.act = act; g.bias = bias; g.out = out; g.ldc = Cout; g.add = add; g.ldadd = Cout;
    g.pixW = Ww; g.pixH = Hh;
    if (uv) g.uv = *uv;
    if (fused) *fused = false;
    if (side && std::is_same<T, f16>::value && Cin == Cout && moge_tune_get("CONV_PP", 1) && moge_tune_get("FUSE_IN", 1)) {
        // fused 1x1 side input (x + in_l(neck_l), modules.py:245): only the halo kernel implements it; `bias` must already hold
        // the sum of both biases (the caller passes the combined vector when *fused comes back true, see below)
        GemmArgs g2 = g;
        g2.a2 = side; g2.w2 = side_w;
        if (conv_pp_eligible(g2)) {
            *fused = true;
            return run_gemm<T>(h, g2, AMODE_CONV3, MOGE_KC_CONV, st, 9.0 * Cin + Cin);
        }
    }
    return run_gemm<T>(h, g, AMODE_CONV3, MOGE_KC_CONV, st);
}

// bilinear x2 + 3x3 conv as a 4-phase 3x3 conv on the low-res map (Hl,Wl), output (B,2Hl,2Wl,Cout); uv given for the HIGH-res grid
template <typename T>
static int conv_up2_phase(moge_handle* h, const T* in, const T* w4, const float* bias4, T* out, int B, int Hl, int Wl, int Cin, int Cout,
                          const UVTerm* uv, hipStream_t st, const float* dot_tab = nullptr, int dot_nd = 0, float* dot_out = nullptr) {
    GemmArgs g = gemm_args();
    g.a = in; g.H = Hl; g.W = Wl; g.C = Cin;
    g.w = w4; g.ldw = 9 * Cin;
    g.M = B * Hl * Wl; g.N = 4 * Cout; g.K = 9 * Cin;
    g.epi = EPI_CONVT; g.bias = bias4; g.out = out; g.Cout = Cout; g.pixW = Wl; g.pixH = Hl;
    if (uv) g.uv = *uv;
    if (dot_tab) { g.dot_tab = dot_tab; g.dot_nd = dot_nd; g.dot_out = dot_out; g.out = nullptr; }      // fused output conv: only (B,2H,2W,4 nd) fp32 leaves the kernel
    return run_gemm<T>(h, g, AMODE_CONV3, MOGE_KC_CONV, st);
}

template <typename T>
static int conv1x1(moge_handle* h, const T* in, const T* w, const float* bias, T* out, long Mrows, int Cin, int Cout, const T* add,
                   const UVTerm* uv, int pixW, int pixH, hipStream_t st) {
    GemmArgs g = gemm_args();
    g.a = in; g.lda = Cin; g.w = w; g.ldw = Cin;
    g.M = (int)Mrows; g.N = Cout; g.K = Cin;
    g.epi = EPI_STORE; g.bias = bias; g.out = out; g.ldc = Cout; g.add = add; g.ldadd = Cout;
    g.pixW = pixW; g.pixH = pixH;
    if (uv) g.uv = *uv;
    return run_gemm<T>(h, g, AMODE_LINEAR, MOGE_KC_CONV, st);
}

template <typename T>
static int convT2(moge_handle* h, const T* in, const T* w, const float* biasT, T* out, int B, int Hh, int Ww, int Cin, int Cout, hipStream_t st) {
    GemmArgs g = gemm_args();
    g.a = in; g.lda = Cin; g.w = w; g.ldw = Cin;
    g.M = B * Hh * Ww; g.N = 4 * Cout; g.K = Cin;
    g.epi = EPI_CONVT; g.bias = biasT; g.out = out; g.Cout = Cout; g.pixW = Ww; g.pixH = Hh;
    return run_gemm<T>(h, g, AMODE_LINEAR, MOGE_KC_CONV, st);
}

// ConvTranspose2d(k2, s2) + 3x3 conv (modules.py:160-165) as ONE composed conv on the low-res map (conv_pp.hip CT3; fp16 path, Cin = 2 Cout) + its border ring:
// in (B, Hl, Wl, Cin) -> out (B, 2 Hl, 2 Wl, Cout); uv at the HIGH-res grid; side = high-res map of the fused 1x1 input block (heads).  *done = false: shape not taken
// (nothing ran).  Profiler: the ALGORITHMIC work of the pair (2 Cin Cout + 9 Cout^2 [+ Cout^2] MACs per output pixel); the kernel executes 16 Cin Cout / 4 per pixel.
template <typename T>
static int convT_conv3_fused(moge_handle* h, const T* in, const T* wc, const T* dw, const float* bias4, T* out, int B, int Hl, int Wl, int Cin, int Cout,
                             const UVTerm* uv, const T* side, const T* side_w, hipStream_t st, bool* done) {
    *done = false;
    if (!std::is_same<T, f16>::value || !moge_tune_get("FUSE_CT3", 1) || !moge_tune_get("CONV_PP", 1)) return 0;
    GemmArgs g = gemm_args();
    g.a = in; g.H = Hl; g.W = Wl; g.C = Cin; g.w = wc; g.ldw = 4 * Cin;
    g.M = B * Hl * Wl; g.N = 4 * Cout; g.K = 4 * Cin;
    g.epi = EPI_CONVT; g.bias = bias4; g.out = out; g.Cout = Cout; g.pixW = Wl; g.pixH = Hl; g.ct3 = 1;
    if (uv) g.uv = *uv;
    if (side) { g.a2 = side; g.w2 = side_w; }
    if (!conv_pp_eligible(g)) return 0;
    const double kalgo = ((2.0 * Cin * Cout + 9.0 * Cout * Cout) * 4 + (side ? 4.0 * Cout * Cout : 0.0)) / (4.0 * Cout);      // flops = 2 M N kalgo
    {
        const double flops = 2.0 * g.M * (double)g.N * kalgo;
        const double bytes = ((double)g.N * g.K + (double)g.M * Cin + (double)g.M * 4 * Cout * (side ? 2 : 1)) * sizeof(T);
        ProfScope ps(h, st, MOGE_KC_CONV, flops, bytes);
        LCHK(launch_conv_pp(g, st));
        LCHK(launch_ct3_border(in, dw, out, B, Hl, Wl, Cin, Cout, st));
    }
    *done = true;
    return 0;
}

// n residual blocks x = x + conv2(relu(conv1(relu(x))))   (modules.py:47-68 with norms = Identity) on x, tmp = scratch of the same size.
// Returns the result in *res: x, or tmp when `may_swap` and an odd number of blocks ran FUSED - the fused kernel (tools/experiments/conv_rb.hip, -DMOGE_EXPERIMENTS builds only: both convs in one
// launch, the intermediate map never leaves LDS) cannot run in place, a neighbouring tile still needs the input pixels it would overwrite.
template <typename T>
static int res_blocks(moge_handle* h, const std::string& name, int l, int n, T* x, T* tmp, int B, int Hh, int Ww, int C, hipStream_t st, bool may_swap = false,
                      T** res = nullptr, T* tmp2 = nullptr, float* gn = nullptr) {
    T* cur = x; T* oth = tmp;
    const bool neck = name == "neck";
    const int in_norm = neck ? h->cfg.neck_in_norm : h->cfg.head_in_norm, hid_norm = neck ? h->cfg.neck_hidden_norm : h->cfg.head_hidden_norm;
    const int act = stack_act(h->cfg, neck), Ch = C * stack_mult(h->cfg, neck);
    for (int j = 0; j < n; j++) {
        const T* w1 = P<T>(h, name + S(".res%d.%d.w1", l, j)); const T* w2 = P<T>(h, name + S(".res%d.%d.w2", l, j));
        const float* b1 = M(h, name + S(".res_blocks.%d.%d.layers.2.bias", l, j)); const float* b2 = M(h, name + S(".res_blocks.%d.%d.layers.5.bias", l, j));
        if (in_norm || hid_norm || act != MOGE_ACT_RELU) {
            // generic block (modules.py:47-67): [norm ->] act -> 3x3 (C -> Ch) -> [norm ->] act -> 3x3 (Ch -> C), + x.  "layer_norm" = GroupNorm(1, C),
            // "group_norm" = GroupNorm(C / 32, C), "instance_norm" = InstanceNorm2d (per channel, no affine); the norm + activation pairs run on
            // MoGe-1's deterministic slab kernels (elementwise.hip).  ReLU without a norm stays fused in the conv (input side / epilogue).
            // (the second scratch map is only touched when the first norm / activation pass writes `oth`; a hidden norm alone is applied in place)
            if ((!tmp2 && (in_norm || act != MOGE_ACT_RELU)) || (!gn && (in_norm || hid_norm)))
                return fail(MOGE_ERR_INVALID, "res_blocks: generic blocks need a second scratch map (and the norm statistics scratch)");
            const std::string r = name + S(".res_blocks.%d.%d.layers.", l, j);
            auto affine = [&](int mode, const char* key) -> const float* { return (mode == MOGE_NORM_LAYER || mode == MOGE_NORM_GROUP) ? M(h, r + key) : nullptr; };
            const T* in1 = cur;
            int relu1 = 1;
            if (in_norm || act != MOGE_ACT_RELU) {
                ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)B * Hh * Ww * C * 2 * sizeof(T));
                LCHK(launch_groupnorm_act<T>(cur, oth, affine(in_norm, "0.weight"), affine(in_norm, "0.bias"), gn, B, Hh, Ww, C, in_norm ? norm_groups(in_norm, C) : 0, act, st));
                in1 = oth; relu1 = 0;
            }
            T* h1 = in1 == oth ? tmp2 : oth;              // conv1's output (Ch channels)
            const bool relu_fused = !hid_norm && act == MOGE_ACT_RELU;
            CHK(conv3x3<T>(h, in1, w1, b1, h1, B, Hh, Ww, C, Ch, relu1, relu_fused ? ACT_RELU : ACT_NONE, nullptr, nullptr, st));
            if (!relu_fused) {
                // elementwise after the statistics pass: in place
                ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)B * Hh * Ww * Ch * 2 * sizeof(T));
                LCHK(launch_groupnorm_act<T>(h1, h1, affine(hid_norm, "3.weight"), affine(hid_norm, "3.bias"), gn, B, Hh, Ww, Ch, hid_norm ? norm_groups(hid_norm, Ch) : 0, act, st));
            }
            CHK(conv3x3<T>(h, h1, w2, b2, cur, B, Hh, Ww, Ch, C, 0, ACT_NONE, cur, nullptr, st));
            continue;
        }
#ifdef MOGE_EXPERIMENTS   // tools/experiments/conv_rb.hip: not part of the product library (python -m moge_amd.build --experiments)
        // CONV_RB (default OFF): measured on MI355X the fused launch is 5-10 % SLOWER than the two conv_pp launches at the bench's level-3 shape
        // (1.44-1.46 vs 1.37-1.38 ms at batch 32, profiles/r03a_kbench_rb.log, timeline r03j: one 130 KiB workgroup per CU exposes every
        // latency the two-per-CU conv_pp form hides, and the MFMA segments run at 21.7 clocks per MFMA beside the partner's read segment)
        if (std::is_same<T, f16>::value && Ch == C && moge_tune_get("CONV_RB", 0) && (may_swap || ((n - j) >= 2) || cur != x)) {
            // (without may_swap the result must end in x: fuse blocks in pairs, or the last one when the data currently sits in tmp)
            GemmArgs g = gemm_args();
            g.a = cur; g.H = Hh; g.W = Ww; g.C = C; g.relu_in = 1; g.w = w1; g.ldw = 9 * C; g.M = B * Hh * Ww; g.N = C; g.K = 9 * C;
            g.epi = EPI_STORE; g.act = ACT_NONE; g.bias = b1; g.out = oth; g.ldc = C; g.add = cur; g.ldadd = C; g.pixW = Ww; g.pixH = Hh;
            g.rb_w2 = w2; g.rb_bias2 = b2;
            if (conv_rb_eligible(g)) {
                // algorithmic work = the two convs; bytes = input map + both weight sets + output map (the skip rows are the input map again, from L2)
                ProfScope ps(h, st, MOGE_KC_CONV, 2.0 * 2.0 * g.M * (double)C * 9.0 * C, (2.0 * g.M * C + 2.0 * 9.0 * C * C) * sizeof(T));
                LCHK(launch_conv_rb(g, st));
                std::swap(cur, oth);
                continue;
            }
        }
#endif
        if (cur != x) return fail(MOGE_ERR_INVALID, "res_blocks: internal buffer order");      // (unreachable: an unfused block only follows an even number of fused ones)
        CHK(conv3x3<T>(h, cur, w1, b1, oth, B, Hh, Ww, C, Ch, 1, ACT_RELU, nullptr, nullptr, st));
        CHK(conv3x3<T>(h, oth, w2, b2, cur, B, Hh, Ww, Ch, C, 0, ACT_NONE, cur, nullptr, st));
    }
    if (res) *res = cur;
    else if (cur != x) return fail(MOGE_ERR_INVALID, "res_blocks: result left in the scratch buffer");
    return 0;
}

// Position embedding of a token grid: computed once per grid and kept in a small LRU (a CLI / evaluation run sees a new grid for every
// aspect ratio: 15 MB per entry for ViT-L at 3600 tokens must not accumulate).  Most recently used entry sits at the back.
static const size_t POS_CACHE_MAX = 8;
static int get_pos(moge_handle* h, int rows, int cols, hipStream_t st, const float** out) {
    for (size_t i = 0; i < h->pos_cache.size(); i++)
        if (h->pos_cache[i].rows == rows && h->pos_cache[i].cols == cols && h->pos_cache[i].mode == h->onnx_mode) {
            const moge_handle::PosEntry e = h->pos_cache[i];
            h->pos_cache.erase(h->pos_cache.begin() + i);
            h->pos_cache.push_back(e);
            *out = e.ptr;
            return 0;
        }
    if (h->pos_cache.size() >= POS_CACHE_MAX) {
        // the evicted buffer may still be read by kernels in flight on this or the split streams: drain the device before freeing it
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipFree(h->pos_cache.front().ptr));
        h->pos_cache.erase(h->pos_cache.begin());
    }
    float* p;
    HIPCHK(hipMalloc(&p, (size_t)(1 + rows * cols) * h->cfg.embed_dim * sizeof(float)));
    LCHK(launch_posembed(M(h, h->bb + "pos_embed"), p, h->cfg.embed_dim, rows, cols, h->onnx_mode, st));
    h->pos_cache.push_back({rows, cols, h->onnx_mode, p});
    *out = p;
    return 0;
}

// ViT encoder shared by both model versions: resize + normalise + patchify of `image` (B, 3, imgH, imgW), patch embedding, the blocks, the
// final-norm taps and their summed 1x1 projections -> pl.feat (B, rows, cols, c0) [+ pl.cls].  v2: modules.py:120-136; v1: v1.py:280-283 +
// the `projects` of Head.forward (v1.py:108-111).
template <typename T>
static int encode(moge_handle* h, const void* image, int img_dtype, int imgH, int imgW, const Plan& pl, hipStream_t st, bool want_feat = true) {
    const moge_config& c = h->cfg;
    const int D = c.embed_dim, nh = c.num_heads, L = c.depth, c0 = c.dims[0];
    const int B = pl.B, rows = pl.rows, cols = pl.cols, Np = pl.Np, Ntok = pl.Ntok, Npad = pl.Npad;
    char* ws = h->ws + pl.base;
    T* patches = (T*)(ws + pl.patches);
    float* x = (float*)(ws + pl.x);
    T* xn = (T*)(ws + pl.xn);
    T* qb = (T*)(ws + pl.q);
    T* kb = (T*)(ws + pl.k);
    T* vT = (T*)(ws + pl.vT);
    T* attn = (T*)(ws + pl.attn);
    T* hidden = (T*)(ws + pl.hidden);
    T* tapcat = (T*)(ws + pl.tapcat);
    float* cls = (float*)(ws + pl.cls);
    T* feat = (T*)(ws + pl.feat);
    const std::string bb = h->bb;
    const long BN = (long)B * Ntok, BP = (long)B * Np;

    // ---- K0: resize + normalise + patchify (modules.py:121-122, patch_embed.py:75) ------------------------------
    const float* mean = h->img_mean; const float* sd = h->img_std;
    {
        ProfScope ps(h, st, MOGE_KC_PRE, 0, (double)B * 3 * imgH * imgW * (img_dtype == 1 ? 2 : 4) + (double)BP * KPATCH_PAD * sizeof(T));
        // Round 6 (one image): preprocess_kernel also zeroes the K padding columns of `patches` (zero_cols_kernel before) and the forward's device-side
        // counters (a hipMemsetAsync = two fill kernels before), and the patch-embed epilogue writes the cls rows (cls_row_kernel before): four launches
        // fewer in front of the first block, the same bits.  (--experiments builds with the stream-K attention workspace: its counters sit behind the LN
        // counters, one range.)
        const bool attn_pp0 = std::is_same<T, f16>::value && moge_tune_get("ATTN_PP", 1) != 0;
        const bool has_attn_ws = attn_pp0 && pl.attn_ws_bytes;
        int* zero_p = pl.ln_cnt_bytes ? (int*)(ws + pl.ln_cnt) : (has_attn_ws ? (int*)(ws + pl.attn_ws) : nullptr);
        const size_t zero_bytes = pl.ln_cnt_bytes ? (has_attn_ws ? (pl.attn_ws - pl.ln_cnt) + attention_pp_ws_counter_bytes(B, nh, Ntok) : pl.ln_cnt_bytes)
                                                  : (has_attn_ws ? attention_pp_ws_counter_bytes(B, nh, Ntok) : 0);
        // img_dtype 3 = fp32 values to be rounded to fp16 on load (the model-dtype cast of a .half() model, v2.py:229)
        if (img_dtype == 0 || img_dtype == 3) LCHK((launch_preprocess<float, T>(image, patches, B, imgH, imgW, rows, cols, KPATCH_PAD, 0, img_dtype == 3, !h->onnx_mode, mean, sd, st, zero_p, (int)((zero_bytes + 3) / 4))));
        else LCHK((launch_preprocess<f16, T>(image, patches, B, imgH, imgW, rows, cols, KPATCH_PAD, 0, 0, !h->onnx_mode, mean, sd, st, zero_p, (int)((zero_bytes + 3) / 4))));
    }
    const float* pos;
    CHK(get_pos(h, rows, cols, st, &pos));
    {   // patch embed GEMM, epilogue adds bias + position embedding and writes the fp32 residual stream (and the cls row of every image)
        GemmArgs g = gemm_args();
        g.a = patches; g.lda = KPATCH_PAD; g.w = P<T>(h, "patch.w"); g.ldw = KPATCH_PAD;
        g.M = (int)BP; g.N = D; g.K = KPATCH_PAD;
        g.epi = EPI_PATCH; g.bias = M(h, bb + "patch_embed.proj.bias"); g.xres = x; g.pos = pos; g.Np = Np; g.Ntok = Ntok;
        g.cls = M(h, bb + "cls_token");
        CHK(run_gemm<T>(h, g, AMODE_LINEAR, MOGE_KC_GEMM, st, KPATCH));
    }
    // fp16 throughput path: V row-major + attention_pp (LDS-DMA, transposed LDS reads); fp32 parity path: V^T + attention.hip
    const bool attn_pp = std::is_same<T, f16>::value && moge_tune_get("ATTN_PP", 1) != 0;
    if (!attn_pp) HIPCHK(hipMemsetAsync(vT, 0, (size_t)B * D * Npad * sizeof(T), st));      // zero the key padding of V^T
    // stream-K attention (--experiments builds; one image: 464 workgroups on 768 slots): its per-query-block counters start at zero (preprocess_kernel above); the kernel leaves them zero
    void* attn_ws = attn_pp && pl.attn_ws_bytes ? (void*)(ws + pl.attn_ws) : nullptr;
    int* ln_cnt = pl.ln_cnt_bytes ? (int*)(ws + pl.ln_cnt) : nullptr;       // fused LN finalize (latency-regime GEMMs): row-block counters, zeroed by preprocess_kernel

    // ---- ViT blocks (block.py:110-112) -----------------------------------------------------------------------------
    // LN fold (fp16 path): norm1 / norm2 never run as kernels.  LN(x) W^T + b = rstd (x W'^T - mean c) + b' with W' = g (.) W: the qkv and
    // fc1 GEMMs read the fp16 COPY of the raw residual (written by the previous RESID epilogue next to its (sum, sum of squares) partials)
    // and apply (mean, rstd) in their epilogues.  Removes 2 x 0.7 GB of LayerNorm traffic per block; the final-norm taps still run
    // layernorm_kernel on the fp32 residual.  The statistics' summation tree is the same in gemm.hip and gemm_pp.hip (batch invariance).
    const bool ln_fold = std::is_same<T, f16>::value && (D % 64) == 0 && moge_tune_get("LN_FOLD", 1) != 0;
    // `.half()` models (MOGE_FP16_HALF): the residual stream lives in `xn` as fp16 from the first block on - the proj / fc2 epilogues update it in
    // place (EPK_RESID16) and the fp32 stream `x` is only the patch-embedding output.  Needs the LN fold (its operand IS the raw stream).
    const bool half_resid = ln_fold && h->half_resid && moge_tune_get("HALF_RESID", 1) != 0;
    float* ln_part = (float*)(ws + pl.ln_part);
    float* ln_mr = (float*)(ws + pl.ln_mr);
    int tap_k = 0;
    bool ln_done = false;          // the producer GEMM in front has written (mean, rstd) itself: no ln_finalize launch
    for (int i = 0; i < L; i++) {
        const std::string p = bb + S("blocks.%d.", i);
        if (!ln_fold) {
            ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)BN * D * (4 + sizeof(T)));
            LCHK(launch_layernorm<T>(x, M(h, p + "norm1.weight"), M(h, p + "norm1.bias"), xn, nullptr, BN, D, D, 0, 0, Ntok, st));
        } else if (i == 0) {
            ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)BN * D * (4 + sizeof(T)));
            LCHK(launch_ln_raw<f16>(x, xn, ln_mr, BN, D, st));          // tokens0 come from the patch-embed epilogue: copy + statistics
        } else if (!ln_done) {
            ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)BN * (D / 32) * 8);
            LCHK(launch_ln_finalize(ln_part, ln_mr, BN, D / 32, D, st));   // partials of block i-1's fc2 epilogue
        }
        {
            GemmArgs g = gemm_args();
            g.a = xn; g.lda = D; g.w = P<T>(h, S(ln_fold ? "blk%d.qkvf" : "blk%d.qkv", i)); g.ldw = D;
            g.M = (int)BN; g.N = 3 * D; g.K = D;
            g.epi = EPI_QKV; g.bias = M(h, p + "attn.qkv.bias"); g.q = qb; g.k = kb; g.vT = vT;
            if (ln_fold) { g.bias = A(h, S("blk%d.qkv.bf", i)); g.ln_mr = ln_mr; g.ln_c = A(h, S("blk%d.qkv.c", i)); }
            g.nh = nh; g.Npad = Npad; g.D = D; g.Ntok = Ntok; g.v_rowmajor = attn_pp ? 1 : 0;
            g.qscale = 0.125f * 1.4426950408889634f;       // 1/sqrt(64) * log2(e): attention works in exp2
            CHK(run_gemm<T>(h, g, AMODE_LINEAR, MOGE_KC_GEMM, st));
        }
        {
            ProfScope ps(h, st, MOGE_KC_ATTN, 4.0 * B * nh * (double)Ntok * Ntok * 64, (double)BN * D * 4 * sizeof(T));
            if (attn_pp) LCHK(launch_attention_pp(qb, kb, vT, attn, B, nh, Ntok, st, attn_ws, pl.attn_ws_bytes));
            else LCHK(launch_attention<T>(qb, kb, vT, attn, B, nh, Ntok, Npad, st));
        }
        {
            GemmArgs g = gemm_args();
            g.a = attn; g.lda = D; g.w = P<T>(h, S("blk%d.proj", i)); g.ldw = D;
            g.M = (int)BN; g.N = D; g.K = D;
            g.epi = EPI_RESID; g.bias = M(h, p + "attn.proj.bias"); g.xres = x; g.ldc = D; g.gamma = M(h, p + "ls1.gamma");
            if (ln_fold) { g.x16 = xn; g.ln_part = ln_part; }
            if (half_resid) g.xres = nullptr;
            ln_done = ln_fold && ln_cnt && gemm_fuses_ln_finalize(g);     // latency regime: the GEMM's last column tile of a row block writes (mean, rstd)
            if (ln_done) { g.ln_mr_out = ln_mr; g.ln_cnt = ln_cnt; }
            CHK(run_gemm<T>(h, g, AMODE_LINEAR, MOGE_KC_GEMM, st));
        }
        if (!ln_fold) {
            ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)BN * D * (4 + sizeof(T)));
            LCHK(launch_layernorm<T>(x, M(h, p + "norm2.weight"), M(h, p + "norm2.bias"), xn, nullptr, BN, D, D, 0, 0, Ntok, st));
        } else if (!ln_done) {
            ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)BN * (D / 32) * 8);
            LCHK(launch_ln_finalize(ln_part, ln_mr, BN, D / 32, D, st));
        }
        {
            GemmArgs g = gemm_args();
            g.a = xn; g.lda = D; g.w = P<T>(h, S(ln_fold ? "blk%d.fc1f" : "blk%d.fc1", i)); g.ldw = D;
            g.M = (int)BN; g.N = 4 * D; g.K = D;
            g.epi = EPI_STORE; g.act = ACT_GELU; g.bias = M(h, p + "mlp.fc1.bias"); g.out = hidden; g.ldc = 4 * D;
            if (ln_fold) { g.bias = A(h, S("blk%d.fc1.bf", i)); g.ln_mr = ln_mr; g.ln_c = A(h, S("blk%d.fc1.c", i)); }
            CHK(run_gemm<T>(h, g, AMODE_LINEAR, MOGE_KC_GEMM, st));
        }
        {
            GemmArgs g = gemm_args();
            g.a = hidden; g.lda = 4 * D; g.w = P<T>(h, S("blk%d.fc2", i)); g.ldw = 4 * D;
            g.M = (int)BN; g.N = D; g.K = 4 * D;
            g.epi = EPI_RESID; g.bias = M(h, p + "mlp.fc2.bias"); g.xres = x; g.ldc = D; g.gamma = M(h, p + "ls2.gamma");
            if (ln_fold && i + 1 < L) { g.x16 = xn; g.ln_part = ln_part; }
            if (half_resid) { g.xres = nullptr; g.x16 = xn; }          // (last block: no statistics wanted, the stream is still updated)
            ln_done = ln_fold && ln_cnt && gemm_fuses_ln_finalize(g);     // (block i + 1's qkv reads ln_mr)
            if (ln_done) { g.ln_mr_out = ln_mr; g.ln_cnt = ln_cnt; }
            CHK(run_gemm<T>(h, g, AMODE_LINEAR, MOGE_KC_GEMM, st));
        }
        for (int k = 0; k < c.n_taps; k++)
            if (c.taps[k] == i) {
                // shared final LayerNorm on the tap, cls/patch split (vision_transformer.py:321-324); cls of the LAST tap only
                // (on a side stream beside the next block's qkv GEMM and attention, which only read the residual stream: measured level at one image,
                //  profiles/r06t_ab_TAP_STREAM_b1.log - the two cross-stream events cost what the 13 us per tap save)
                ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)BN * D * ((half_resid ? 2 : 4) + sizeof(T)));
                if (half_resid)
                    LCHK(launch_layernorm_x16(xn, M(h, bb + "norm.weight"), M(h, bb + "norm.bias"), tapcat, k == c.n_taps - 1 ? cls : nullptr, BN, D,
                                              c.n_taps * D, k * D, 1, Ntok, st));
                else
                LCHK(launch_layernorm<T>(x, M(h, bb + "norm.weight"), M(h, bb + "norm.bias"), tapcat, k == c.n_taps - 1 ? cls : nullptr, BN, D,
                                         c.n_taps * D, k * D, 1, Ntok, st));
                tap_k++;
            }
    }
    if (tap_k != c.n_taps) return fail(MOGE_ERR_INVALID, "intermediate_layers must be distinct block indices < depth");
    if (want_feat) {   // sum_k Conv1x1_k(tap_k) == one GEMM over K = n_taps*D (modules.py:128-131)
        GemmArgs g = gemm_args();
        g.a = tapcat; g.lda = c.n_taps * D; g.w = P<T>(h, "outproj.w"); g.ldw = c.n_taps * D;
        g.M = (int)BP; g.N = c0; g.K = c.n_taps * D;
        g.epi = EPI_STORE; g.bias = A(h, "outproj.bias"); g.out = feat; g.ldc = c0;
        CHK(run_gemm<T>(h, g, AMODE_LINEAR, MOGE_KC_GEMM, st));
    }
    return 0;
}

template <typename T>
static int forward_impl(moge_handle* h, const void* image, int img_dtype, const Plan& pl, float* o_points, float* o_normal, float* o_maskprob,
                        float* o_metric, hipStream_t st) {
    const moge_config& c = h->cfg;
    const int D = c.embed_dim, c0 = c.dims[0];
    const int B = pl.B, rows = pl.rows, cols = pl.cols, Np = pl.Np, Ntok = pl.Ntok;
    const double aspect = (double)pl.W / (double)pl.H;
    char* ws = h->ws + pl.base;
    float* cls = (float*)(ws + pl.cls);
    T* feat = (T*)(ws + pl.feat);
    const long BN = (long)B * Ntok, BP = (long)B * Np;
    // fp16 path: chains of linear layers with nothing between them are composed at pack time (compose_weights_f16): the summed output
    // projections + the neck's level-0 input block become one GEMM on the K-concatenated taps (`feat` is never formed), and a head without
    // level-0 residual blocks applies (ConvTranspose2d . input block) to the neck's level-0 map in one GEMM.  Same mathematics, fewer
    // roundings; the fp32 parity path keeps the reference's layer-by-layer order.
    const bool compose = std::is_same<T, f16>::value && moge_tune_get("COMPOSE", 1) != 0;
    CHK(encode<T>(h, image, img_dtype, pl.H, pl.W, pl, st, !compose));
    // (the scale head runs behind the heads, see below)

    // fp16 path, level 4 (no residual blocks there, 32 channels): out_k(x4_k + in4_k(n4)) = Wout_k x4_k + (Wout_k Win4_k) n4 + b2_k is evaluated
    // INSIDE the two 64 -> 4 x 32 resampler convs (conv_pp.hip, fused output conv); head_final only resizes and remaps 4 + 4 floats per tap
    int nheads = 0;
    for (int k = 0; k < 3; k++) nheads += (c.heads & HEAD_BITS[k]) ? 1 : 0;
    const bool l4dot = std::is_same<T, f16>::value && c.head_res_blocks[MOGE_LEVELS - 1] == 0 && c.neck_res_blocks[MOGE_LEVELS - 1] == 0 && c.dims[4] == 32 &&
                       c.dims[3] == 64 && rs_is_phase(c.neck_resamplers[3]) && rs_is_phase(c.head_resamplers[3]) &&
                       moge_tune_get("FUSE_L4", 1) != 0 && moge_tune_get("L4DOT", 1) != 0 && moge_tune_get("CONV_PP", 1) != 0;
    // small batches: the heads on their own streams and scratch triples (same kernels: bit-identical); not under the profiler, whose events bracket
    // launches on ONE stream.  HEAD_PIPE (default, round 6): every head on a side stream, level l of a head released by the event of neck level l;
    // HEAD_PIPE 0 (rounds 3-5): the first head on the caller's stream, the others forked behind the whole neck
    const bool par_heads = pl.head_sets > 1 && !h->prof_on;
    const bool pipe = par_heads && moge_tune_get("HEAD_PIPE", 1) != 0;
    auto neck_level_done = [&](int l) -> int {
        if (!pipe) return 0;
        if (!h->ev_lvl[pl.slot][l]) HIPCHK(hipEventCreateWithFlags(&h->ev_lvl[pl.slot][l], hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->ev_lvl[pl.slot][l], st));
        return 0;
    };
    // ---- neck (modules.py:242-254; level-0 uv concat folded into a rank-2 epilogue term, v2.py:154-160) -----------
    T* N[MOGE_LEVELS];
    for (int l = 0; l < MOGE_LEVELS; l++) N[l] = (T*)(ws + pl.neck[l]);
    T* Sc[3] = {(T*)(ws + pl.scratch[0]), (T*)(ws + pl.scratch[1]), (T*)(ws + pl.scratch[2])};
    float* gn = (float*)(ws + pl.gn);                  // GroupNorm partial sums of the normalised residual blocks (ABI v3 options; unused in the released layout)
    {
        UVTerm uv = uv_term(A(h, "neck.in0.wu"), A(h, "neck.in0.wv"), cols, rows, aspect);
        if (compose) {
            GemmArgs g = gemm_args();
            g.a = ws + pl.tapcat; g.lda = c.n_taps * D; g.w = P<T>(h, "neck.in0c.w"); g.ldw = c.n_taps * D;
            g.M = (int)BP; g.N = c0; g.K = c.n_taps * D;
            g.epi = EPI_STORE; g.bias = A(h, "neck.in0c.bias"); g.out = N[0]; g.ldc = c0; g.pixW = cols; g.pixH = rows; g.uv = uv;
            CHK(run_gemm<T>(h, g, AMODE_LINEAR, MOGE_KC_GEMM, st));
        } else
        CHK(conv1x1<T>(h, feat, P<T>(h, "neck.in0.w"), M(h, "neck.input_blocks.0.bias"), N[0], BP, c0, c0, nullptr, &uv, cols, rows, st));
        CHK(res_blocks<T>(h, "neck", 0, c.neck_res_blocks[0], N[0], Sc[0], B, rows, cols, c0, st, false, nullptr, Sc[1], gn));
        CHK(neck_level_done(0));
        for (int l = 1; l < MOGE_LEVELS; l++) {
            const int Hh = rows << l, Ww = cols << l, ci = c.dims[l - 1], co = c.dims[l];
            const int rsl = c.neck_resamplers[l - 1];
            // level l's input block sees the (u, v) planes only (v2.py:154-160): a rank-2 term + bias, carried by the epilogue of the resampler's LAST conv
            UVTerm uvl = uv_term(A(h, S("neck.in%d.wu", l)), A(h, S("neck.in%d.wv", l)), Ww, Hh, aspect);
            if (rsl == MOGE_RS_CONV_TRANSPOSE) {
                bool fused = false;
                if (std::is_same<T, f16>::value && ct3_shape(ci, co))
                    CHK(convT_conv3_fused<T>(h, N[l - 1], P<T>(h, S("neck.rs%d.wc", l - 1)), P<T>(h, S("neck.rs%d.dw", l - 1)), A(h, S("neck.rs%d.bias_ct3", l - 1)), N[l], B,
                                             Hh / 2, Ww / 2, ci, co, &uvl, nullptr, nullptr, st, &fused));
                if (!fused) {
                CHK(convT2<T>(h, N[l - 1], P<T>(h, S("neck.rs%d.wT", l - 1)), A(h, S("neck.rs%d.biasT", l - 1)), Sc[0], B, Hh / 2, Ww / 2, ci, co, st));
                CHK(conv3x3<T>(h, Sc[0], P<T>(h, S("neck.rs%d.w3", l - 1)), A(h, S("neck.rs%d.bias2", l - 1)), N[l], B, Hh, Ww, co, co, 0, ACT_NONE, nullptr,
                               &uvl, st));
                }
            } else if (rsl == MOGE_RS_PIXEL_SHUFFLE) {
                // Conv2d(ci, 4 co) + PixelShuffle = a 3x3 conv on the low-res map stored through the pixel-shuffle epilogue, then the second 3x3 conv
                CHK(conv_up2_phase<T>(h, N[l - 1], P<T>(h, S("neck.rs%d.w3p", l - 1)), A(h, S("neck.rs%d.bias4", l - 1)), Sc[0], B, Hh / 2, Ww / 2, ci, co, nullptr, st));
                CHK(conv3x3<T>(h, Sc[0], P<T>(h, S("neck.rs%d.w3", l - 1)), A(h, S("neck.rs%d.bias2", l - 1)), N[l], B, Hh, Ww, co, co, 0, ACT_NONE, nullptr,
                               &uvl, st));
            } else if (l == MOGE_LEVELS - 1 && l4dot) {
                // the neck's level-4 map is only ever read through the heads' composed input block + output conv (no residual blocks at level
                // 4): the resampler applies all of them per pixel and stores 4 floats per head instead of 32 halves
                CHK(conv_up2_phase<T>(h, N[l - 1], P<T>(h, "neck.rs3.w3p"), A(h, "neck.rs3.bias4"), N[l], B, Hh / 2, Ww / 2, ci, co, &uvl, st, A(h, "neck.dot"), nheads,
                                      (float*)N[l]));
            } else {
                CHK(conv_up2_phase<T>(h, N[l - 1], P<T>(h, S("neck.rs%d.w3p", l - 1)), A(h, S("neck.rs%d.bias4", l - 1)), N[l], B, Hh / 2, Ww / 2, ci, co, &uvl, st));
            }
            CHK(res_blocks<T>(h, "neck", l, c.neck_res_blocks[l], N[l], Sc[1], B, Hh, Ww, co, st, false, nullptr, Sc[2], gn));
            CHK(neck_level_done(l));
        }
    }
    // ---- heads -------------------------------------------------------------------------------------------------------
    float* outs[3] = {o_points, o_normal, o_maskprob};
    const bool fuse_l4 = std::is_same<T, f16>::value && c.head_res_blocks[MOGE_LEVELS - 1] == 0 && moge_tune_get("FUSE_L4", 1) != 0;
    hipStream_t st_main = st;
    int forked = 0;
    // Whatever way this function is left (a failed launch in the middle of a head included), the caller's stream must be ordered behind every
    // head stream that was forked: their kernels use the shared workspace and the caller's output buffers, which a later ensure_ws() / hipFree
    // or the caller itself may touch as soon as `st_main` looks idle.
    struct HeadJoin {
        moge_handle* h; int slot; hipStream_t main; int* forked;
        ~HeadJoin() {
            for (int i = 0; i < 3; i++)
                if (*forked & (1 << i)) {
                    if (hipEventRecord(h->ev_head[slot][i], h->head_st[slot][i]) == hipSuccess) hipStreamWaitEvent(main, h->ev_head[slot][i], 0);
                }
            *forked = 0;
        }
    } head_join{h, pl.slot, st_main, &forked};
    if (par_heads && !pipe) {
        if (!h->ev_neck[pl.slot]) HIPCHK(hipEventCreateWithFlags(&h->ev_neck[pl.slot], hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->ev_neck[pl.slot], st_main));
    }
    for (int k = 0; k < 3; k++) {
        if (!(c.heads & HEAD_BITS[k]) || !outs[k]) continue;
        const std::string name = HEAD_NAMES[k];
        int head_idx = 0;                  // this head's group in the neck's fused-output-conv table: its position among the model's heads
        for (int k2 = 0; k2 < k; k2++) head_idx += (c.heads & HEAD_BITS[k2]) ? 1 : 0;
        if (par_heads && (pipe || head_idx > 0)) {
            const int sidx = pipe ? head_idx : head_idx - 1;           // side stream / join event of this head
            hipStream_t& hs = h->head_st[pl.slot][sidx];
            if (!hs) {
                HIPCHK(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&h->ev_head[pl.slot][sidx], hipEventDisableTiming));
            }
            HIPCHK(hipStreamWaitEvent(hs, pipe ? h->ev_lvl[pl.slot][0] : h->ev_neck[pl.slot], 0));
            st = hs;
            for (int i = 0; i < 3; i++) Sc[i] = (T*)(ws + pl.scratch[3 * (pipe ? head_idx + 1 : head_idx) + i]);
            forked |= 1 << sidx;
        } else {
            st = st_main;
            for (int i = 0; i < 3; i++) Sc[i] = (T*)(ws + pl.scratch[i]);
        }
        int cur = 0;                       // Sc[cur] holds the running x
        const bool compose0 = compose && c.head_res_blocks[0] == 0 && c.head_resamplers[0] == MOGE_RS_CONV_TRANSPOSE;       // level 0 -> 1: ConvTranspose2d(input block(n0)) as one GEMM on n0
        if (!compose0) {
            CHK(conv1x1<T>(h, N[0], P<T>(h, name + ".in0.w"), M(h, name + ".input_blocks.0.bias"), Sc[cur], BP, c0, c0, nullptr, nullptr, cols, rows, st));
            CHK(res_blocks<T>(h, name, 0, c.head_res_blocks[0], Sc[cur], Sc[(cur + 1) % 3], B, rows, cols, c0, st, false, nullptr, Sc[(cur + 2) % 3], gn));
        }
        for (int l = 1; l < MOGE_LEVELS; l++) {
            const int Hh = rows << l, Ww = cols << l, ci = c.dims[l - 1], co = c.dims[l];
            const int a = (cur + 1) % 3, b2 = (cur + 2) % 3;
            if (pipe) HIPCHK(hipStreamWaitEvent(st, h->ev_lvl[pl.slot][l], 0));       // this level reads neck level l (side input / level-4 dot products)
            const int rsl = c.head_resamplers[l - 1];
            int nxt;
            bool in_fused = false;
            bool ct3_done = false;
            if (rsl == MOGE_RS_CONV_TRANSPOSE && std::is_same<T, f16>::value && ct3_shape(ci, co) && !(l == 1 && compose0) && moge_tune_get("FUSE_IN", 1) != 0 &&
                !(l == MOGE_LEVELS - 1 && fuse_l4)) {
                // ConvTranspose2d + 3x3 + the head's `x + in_l(neck_l)` (modules.py:160-165, 245) in one composed conv: Sc[cur] (low-res) -> Sc[b2] (high-res)
                CHK(convT_conv3_fused<T>(h, Sc[cur], P<T>(h, name + S(".rs%d.wc", l - 1)), P<T>(h, name + S(".rs%d.dw", l - 1)), A(h, name + S(".rs%d.bias_ct3", l - 1)), Sc[b2], B,
                                         Hh / 2, Ww / 2, ci, co, nullptr, N[l], P<T>(h, name + S(".in%d.w", l)), st, &ct3_done));
                if (ct3_done) { in_fused = true; nxt = b2; }
            }
            if (ct3_done) {
            } else
            if (rsl == MOGE_RS_CONV_TRANSPOSE || rsl == MOGE_RS_PIXEL_SHUFFLE) {
                if (rsl == MOGE_RS_PIXEL_SHUFFLE)
                    CHK(conv_up2_phase<T>(h, Sc[cur], P<T>(h, name + S(".rs%d.w3p", l - 1)), A(h, name + S(".rs%d.bias4", l - 1)), Sc[a], B, Hh / 2, Ww / 2, ci, co, nullptr, st));
                else if (l == 1 && compose0)
                    CHK(convT2<T>(h, N[0], P<T>(h, name + ".rs0.wTc"), A(h, name + ".rs0.biasTc"), Sc[a], B, Hh / 2, Ww / 2, ci, co, st));
                else
                CHK(convT2<T>(h, Sc[cur], P<T>(h, name + S(".rs%d.wT", l - 1)), A(h, name + S(".rs%d.biasT", l - 1)), Sc[a], B, Hh / 2, Ww / 2, ci, co, st));
                const float* plain_bias = M(h, name + S(".resamplers.%d.%s.bias", l - 1, rs_final_conv(rsl)));
                // resampler conv, with the head's `x + in_l(neck_l)` fused as a 1x1 side input when the halo kernel takes the shape
                // (the first attempt passes the combined bias; if the shape is not eligible nothing ran and the plain form follows)
                const bool try_fuse = std::is_same<T, f16>::value && rsl == MOGE_RS_CONV_TRANSPOSE && moge_tune_get("FUSE_IN", 1) != 0 && moge_tune_get("CONV_PP", 1) != 0 &&
                                      !(l == MOGE_LEVELS - 1 && fuse_l4);
                if (try_fuse) {
                    GemmArgs probe = gemm_args();
                    probe.a = Sc[a]; probe.H = Hh; probe.W = Ww; probe.C = co; probe.w = P<T>(h, name + S(".rs%d.w3", l - 1)); probe.ldw = 9 * co;
                    probe.M = B * Hh * Ww; probe.N = co; probe.K = 9 * co; probe.epi = EPI_STORE; probe.bias = A(h, name + S(".rs%d.bias_in", l - 1));
                    probe.out = Sc[b2]; probe.ldc = co; probe.a2 = N[l]; probe.w2 = P<T>(h, name + S(".in%d.w", l));
                    if (conv_pp_eligible(probe)) {
                        CHK(conv3x3<T>(h, Sc[a], P<T>(h, name + S(".rs%d.w3", l - 1)), A(h, name + S(".rs%d.bias_in", l - 1)), Sc[b2], B, Hh, Ww, co, co, 0,
                                       ACT_NONE, nullptr, nullptr, st, N[l], P<T>(h, name + S(".in%d.w", l)), &in_fused));
                    }
                }
                if (!in_fused)
                    CHK(conv3x3<T>(h, Sc[a], P<T>(h, name + S(".rs%d.w3", l - 1)), plain_bias, Sc[b2], B, Hh, Ww, co, co, 0,
                                   ACT_NONE, nullptr, nullptr, st));
                nxt = b2;
            } else if (l == MOGE_LEVELS - 1 && l4dot) {
                CHK(conv_up2_phase<T>(h, Sc[cur], P<T>(h, name + ".rs3.w3p"), A(h, name + ".rs3.bias4"), Sc[a], B, Hh / 2, Ww / 2, ci, co, nullptr, st, A(h, name + ".dot.own"), 1,
                                      (float*)Sc[a]));
                nxt = a;
            } else {
                CHK(conv_up2_phase<T>(h, Sc[cur], P<T>(h, name + S(".rs%d.w3p", l - 1)), A(h, name + S(".rs%d.bias4", l - 1)), Sc[a], B, Hh / 2, Ww / 2, ci, co, nullptr, st));
                nxt = a;
            }
            // x = x + in_l(neck_l)   (in place: each element is read and written by the same lane); at the last level of the fp16
            // path the input block is folded into the output conv of head_final_kernel (no residual blocks in between)
            if (!(l == MOGE_LEVELS - 1 && fuse_l4) && !in_fused)
                CHK(conv1x1<T>(h, N[l], P<T>(h, name + S(".in%d.w", l)), M(h, name + S(".input_blocks.%d.bias", l)), Sc[nxt], (long)B * Hh * Ww, co, co, Sc[nxt],
                               nullptr, Ww, Hh, st));
            cur = nxt;
            {
                T* r = nullptr;                 // fused blocks ping-pong between the two buffers: follow the result
                CHK(res_blocks<T>(h, name, l, c.head_res_blocks[l], Sc[cur], Sc[(cur + 1) % 3], B, Hh, Ww, co, st, true, &r, Sc[(cur + 2) % 3], gn));
                if (r != Sc[cur]) cur = (cur + 1) % 3;
            }
        }
        {
            ProfScope ps(h, st, MOGE_KC_POST, 0, (double)B * (rows << 4) * (cols << 4) * c.dims[4] * sizeof(T));
            if (l4dot)
                LCHK(launch_head_final_dot(k, (const float*)Sc[cur], (const float*)N[4], 4 * nheads, 4 * head_idx, A(h, name + ".out4.b2"), outs[k], B, rows << 4, cols << 4,
                                           pl.H, pl.W, c.remap_output, st));
            else if (fuse_l4)
                LCHK(launch_head_final<T>(k, Sc[cur], M(h, name + ".output_blocks.4.weight"), A(h, name + ".out4.b2"), N[4], A(h, name + ".out4.w2"), outs[k], B,
                                          rows << 4, cols << 4, c.dims[4], pl.H, pl.W, c.remap_output, st));
            else
                LCHK(launch_head_final<T>(k, Sc[cur], M(h, name + ".output_blocks.4.weight"), M(h, name + ".output_blocks.4.bias"), nullptr, nullptr, outs[k], B,
                                          rows << 4, cols << 4, c.dims[4], pl.H, pl.W, c.remap_output, st));
        }
    }
    st = st_main;
    for (int i = 0; i < 3; i++) Sc[i] = (T*)(ws + pl.scratch[i]);
    // ---- scale head (modules.py:184-192, v2.py:167,182) ---------------------------------------------------------
    // Three tiny launches on the caller's stream BEHIND its head: with the other heads on their own streams (small batches) the caller's stream finishes its
    // head first and would idle until the join - in front of the neck (rounds 1-5) the same 15 us sat on the batch-1 critical path.
    if ((c.heads & MOGE_HEAD_SCALE) && o_metric) {
        float* m1 = (float*)(ws + pl.mlp1); float* m2 = (float*)(ws + pl.mlp2);
        ProfScope ps(h, st, MOGE_KC_POST, 0, 0);
        LCHK(launch_mlp_layer(cls, M(h, "scale_head.0.weight"), M(h, "scale_head.0.bias"), m1, B, D, c.scale_hidden, 1, st));
        LCHK(launch_mlp_layer(m1, M(h, "scale_head.2.weight"), M(h, "scale_head.2.bias"), m2, B, c.scale_hidden, c.scale_hidden, 1, st));
        LCHK(launch_mlp_layer(m2, M(h, "scale_head.4.weight"), M(h, "scale_head.4.bias"), o_metric, B, c.scale_hidden, 1, 2, st));
    }
    for (int i = 0; i < 3; i++)
        if (forked & (1 << i)) {
            HIPCHK(hipEventRecord(h->ev_head[pl.slot][i], h->head_st[pl.slot][i]));
            HIPCHK(hipStreamWaitEvent(st_main, h->ev_head[pl.slot][i], 0));
            forked &= ~(1 << i);
        }
    // remember buffers for debug taps
    h->last.valid = true; h->last.prec = TT<T>::PREC; h->last.B = B; h->last.rows = rows; h->last.cols = cols;
    h->last.bufs.clear();
    if (std::is_same<T, f16>::value && h->half_resid && moge_tune_get("HALF_RESID", 1) != 0 && moge_tune_get("LN_FOLD", 1) != 0) h->last.bufs["x_final"] = {pl.xn, {(int64_t)BN * D, 1}};
    else h->last.bufs["x_final"] = {pl.x, {(int64_t)BN * D, 0}};
    h->last.bufs["tapcat"] = {pl.tapcat, {(int64_t)BP * c.n_taps * D, 1}};
    h->last.bufs["cls"] = {pl.cls, {(int64_t)B * D, 0}};
    if (!compose) h->last.bufs["features"] = {pl.feat, {(int64_t)BP * c0, 1}};      // (composed path: never formed)
    for (int l = 0; l < MOGE_LEVELS; l++)
        if (!(l == MOGE_LEVELS - 1 && l4dot))       // (fused output conv: the level-4 map is never materialised)
            h->last.bufs[S("neck%d", l)] = {pl.neck[l], {(int64_t)BP * ((int64_t)1 << (2 * l)) * c.dims[l], 1}};
    return 0;
}


// ============================================================================================================================
// MoGe-1 (moge/model/v1.py): tables, packing, plan, forward.  The ViT part is shared (build_tables / pack_weights /
// encode above); this is the Head (v1.py:61-142) and the two-stage input resize (v1.py:271-278).
// ============================================================================================================================
static int v1_cpad(int c) { return (c + 2 + 7) / 8 * 8; }            // channels + (u, v), padded to 16-byte chunks in both storage types
static int v1_mult(const moge_v1_config& c) { return c.hidden_mult > 0 ? c.hidden_mult : 1; }                          // dim_times_res_block_hidden (v1.py:85)
static int v1_ks(const moge_v1_config& c) { return c.last_conv_size == 3 ? 3 : 1; }
static bool v1_generic_out(const moge_v1_config& c) { return c.last_res_blocks > 0 || v1_ks(c) == 3; }      // output blocks beyond [3x3 -> ReLU -> 1x1] (v1.py:103-109)
static int v1_hidden_groups(const moge_v1_config& c, int ch) { return c.res_block_norm == MOGE_NORM_LAYER ? 1 : (ch / 32 > 0 ? ch / 32 : 1); }     // v1.py:47

static void build_tables_v1_decoder(moge_handle* h) {
    const moge_v1_config& c = h->cfg1;
    for (int i = 0; i < c.n_up; i++) {
        const int ci = i == 0 ? c.dim_proj : c.dim_upsample[i - 1], co = c.dim_upsample[i];
        const std::string u = S("head.upsample_blocks.%d.", i);
        tadd(h, u + "0.0.weight", (int64_t)(ci + 2) * co * 4); tadd(h, u + "0.0.bias", co);
        tadd(h, u + "0.1.weight", (int64_t)co * co * 9); tadd(h, u + "0.1.bias", co);
        padd(h, S("v1.up%d.wT", i), (int64_t)4 * co * ci);
        padd(h, S("v1.up%d.w3", i), (int64_t)co * 9 * co);
        aadd(h, S("v1.up%d.wu", i), 4 * co); aadd(h, S("v1.up%d.wv", i), 4 * co); aadd(h, S("v1.up%d.biasT", i), 4 * co);
        for (int j = 0; j < c.num_res_blocks; j++) {
            const std::string r = u + S("%d.layers.", 1 + j);
            const int ch = co * v1_mult(c);                      // hidden width (v1.py:85)
            tadd(h, r + "0.weight", co); tadd(h, r + "0.bias", co);
            tadd(h, r + "2.weight", (int64_t)ch * co * 9); tadd(h, r + "2.bias", ch);
            tadd(h, r + "3.weight", ch); tadd(h, r + "3.bias", ch);
            tadd(h, r + "5.weight", (int64_t)co * ch * 9); tadd(h, r + "5.bias", co);
            padd(h, S("v1.up%d.res%d.w1", i, j), (int64_t)ch * 9 * co);
            padd(h, S("v1.up%d.res%d.w2", i, j), (int64_t)co * 9 * ch);
        }
    }
    const int cl = c.dim_upsample[c.n_up - 1], c4 = c.last_conv_channels;
    const int nl = c.last_res_blocks, ks = v1_ks(c), ch4 = c4 * v1_mult(c);
    for (int o = 0; o < 2; o++) {
        // nn.Sequential(3x3, ResidualConvBlock x last_res_blocks, ReLU, Conv2d(k = last_conv_size))   (v1.py:103-109)
        const std::string b = S("head.output_block.%d.", o);
        tadd(h, b + "0.weight", (int64_t)c4 * (cl + 2) * 9); tadd(h, b + "0.bias", c4);
        for (int j = 0; j < nl; j++) {
            const std::string r = b + S("%d.layers.", 1 + j);
            tadd(h, r + "0.weight", c4); tadd(h, r + "0.bias", c4);
            tadd(h, r + "2.weight", (int64_t)ch4 * c4 * 9); tadd(h, r + "2.bias", ch4);
            tadd(h, r + "3.weight", ch4); tadd(h, r + "3.bias", ch4);
            tadd(h, r + "5.weight", (int64_t)c4 * ch4 * 9); tadd(h, r + "5.bias", c4);
            padd(h, S("v1.out%d.res%d.w1", o, j), (int64_t)ch4 * 9 * c4);
            padd(h, S("v1.out%d.res%d.w2", o, j), (int64_t)c4 * 9 * ch4);
        }
        tadd(h, b + S("%d.weight", nl + 2), (int64_t)(o == 0 ? 3 : 1) * c4 * ks * ks); tadd(h, b + S("%d.bias", nl + 2), o == 0 ? 3 : 1);
    }
    padd(h, "v1.out.w3", (int64_t)2 * c4 * 9 * v1_cpad(cl));
    aadd(h, "v1.out.bias", 2 * c4);
}

static int build_aux_v1(moge_handle* h, hipStream_t st) {
    const moge_v1_config& c = h->cfg1;
    for (int i = 0; i < c.n_up; i++) {
        const int ci = i == 0 ? c.dim_proj : c.dim_upsample[i - 1], co = c.dim_upsample[i];
        const float* w = M(h, S("head.upsample_blocks.%d.0.0.weight", i));              // [ci + 2][co][2][2]; rows ci, ci + 1 = (u, v) inputs
        LCHK(launch_repack<float>(w + (size_t)ci * co * 4, A(h, S("v1.up%d.wu", i)), 4, co, 1, 1, 1, 4, 0, 0, co, 1, 0, st));
        LCHK(launch_repack<float>(w + (size_t)(ci + 1) * co * 4, A(h, S("v1.up%d.wv", i)), 4, co, 1, 1, 1, 4, 0, 0, co, 1, 0, st));
        LCHK(launch_repack<float>(M(h, S("head.upsample_blocks.%d.0.0.bias", i)), A(h, S("v1.up%d.biasT", i)), 4, 1, 1, co, 0, 0, 0, 1, co, 0, 0, st));
    }
    const int c4 = c.last_conv_channels;
    for (int o = 0; o < 2; o++)
        LCHK(launch_repack<float>(M(h, S("head.output_block.%d.0.bias", o)), A(h, "v1.out.bias") + o * c4, 1, 1, 1, c4, 0, 0, 0, 1, 0, 0, 0, st));
    return 0;
}

template <typename T>
static int pack_weights_v1(moge_handle* h, hipStream_t st) {
    const moge_v1_config& c = h->cfg1;
    auto conv3 = [&](const float* w, T* dst, int co, int ci, int cip) -> int {        // torch [co][ci][3][3] -> [co][tap * cip + ci], cip >= ci zero padded
        return launch_repack<T>(w, dst, co, 9, 1, ci, (long)ci * 9, 1, 0, 9, (long)9 * cip, cip, 0, st);
    };
    for (int i = 0; i < c.n_up; i++) {
        const int ci = i == 0 ? c.dim_proj : c.dim_upsample[i - 1], co = c.dim_upsample[i];
        const std::string u = S("head.upsample_blocks.%d.", i);
        // ConvTranspose2d weight [ci + 2][co][2][2] -> [(dy*2+dx)*co + o][ci] (the two uv input rows go to the epilogue: build_aux_v1)
        LCHK(launch_repack<T>(M(h, u + "0.0.weight"), Pm<T>(h, S("v1.up%d.wT", i)), 4, co, 1, ci, 1, 4, 0, (long)co * 4, (long)co * ci, ci, 0, st));
        LCHK(conv3(M(h, u + "0.1.weight"), Pm<T>(h, S("v1.up%d.w3", i)), co, co, co));
        for (int j = 0; j < c.num_res_blocks; j++) {
            const std::string r = u + S("%d.layers.", 1 + j);
            const int ch = co * v1_mult(c);
            LCHK(conv3(M(h, r + "2.weight"), Pm<T>(h, S("v1.up%d.res%d.w1", i, j)), ch, co, co));
            LCHK(conv3(M(h, r + "5.weight"), Pm<T>(h, S("v1.up%d.res%d.w2", i, j)), co, ch, ch));
        }
    }
    const int cl = c.dim_upsample[c.n_up - 1], c4 = c.last_conv_channels, cp = v1_cpad(cl), ch4 = c4 * v1_mult(c);
    for (int o = 0; o < 2; o++) {
        LCHK(conv3(M(h, S("head.output_block.%d.0.weight", o)), Pm<T>(h, "v1.out.w3") + (size_t)o * c4 * 9 * cp, c4, cl + 2, cp));
        for (int j = 0; j < c.last_res_blocks; j++) {
            const std::string r = S("head.output_block.%d.%d.layers.", o, 1 + j);
            LCHK(conv3(M(h, r + "2.weight"), Pm<T>(h, S("v1.out%d.res%d.w1", o, j)), ch4, c4, c4));
            LCHK(conv3(M(h, r + "5.weight"), Pm<T>(h, S("v1.out%d.res%d.w2", o, j)), c4, ch4, ch4));
        }
    }
    return 0;
}

struct PlanV1 {
    Plan p;                       // encoder buffers + the caller-visible post buffers (same fields as MoGe-2)
    int rh, rw;                   // resized image (v1.py:272-274)
    size_t img1, X, T1, T2, R, Y, gn;
    size_t La, Lb;                // generic output blocks only: norm output (c4) and hidden map (k c4) at the resized image's size
};
static PlanV1 make_plan_v1(moge_handle* h, int prec, int B, int H, int W, int rh, int rw) {
    const moge_config& c = h->cfg;
    const moge_v1_config& c1 = h->cfg1;
    PlanV1 v;
    Plan& p = v.p;
    const size_t s = prec == MOGE_FP16 ? 2 : 4;
    const int D = c.embed_dim;
    const int rows = rh / 14, cols = rw / 14;
    v.rh = rh; v.rw = rw;
    p.B = B; p.H = H; p.W = W; p.rows = rows; p.cols = cols;
    p.Np = rows * cols; p.Ntok = p.Np + 1; p.Npad = (p.Ntok + 63) / 64 * 64;
    const size_t BN = (size_t)B * p.Ntok, BP = (size_t)B * p.Np, px = (size_t)B * H * W;
    p.maskprob = take(p, px * 4);
    p.pts_tmp = take(p, px * 12);
    p.nrm_tmp = take(p, 16);
    p.focal = take(p, (size_t)B * 4); p.shift = take(p, (size_t)B * 4); p.intr = take(p, (size_t)B * 36); p.metric = take(p, (size_t)B * 4);
    p.post_end = p.total;
    p.patches = take(p, BP * KPATCH_PAD * s);
    p.x = take(p, BN * D * 4);
    p.xn = take(p, BN * D * s);
    p.ln_part = take(p, BN * (size_t)(D / 32) * 8);
    p.ln_mr = take(p, BN * 8);
    p.q = take(p, BN * D * s); p.k = take(p, BN * D * s);
    p.vT = take(p, (size_t)B * D * p.Npad * s);
    p.attn = take(p, BN * D * s);
    p.hidden = take(p, BN * 4 * D * s);
    if (prec == MOGE_FP16) {      // (adjacent: one memset zeroes the LN counters and the attention counters at the head of attn_ws)
        p.ln_cnt_bytes = (BN / 64 + 2) * 4; p.ln_cnt = take(p, p.ln_cnt_bytes);
        p.attn_ws_bytes = attention_pp_ws_bytes(B, c.num_heads, p.Ntok); p.attn_ws = take(p, p.attn_ws_bytes);
    }
    p.tapcat = take(p, BP * c.n_taps * D * s);
    p.cls = take(p, (size_t)B * D * 4);
    p.mlp1 = p.mlp2 = 0;
    p.feat = take(p, BP * c.dims[0] * s);
    for (int l = 0; l < MOGE_LEVELS; l++) p.neck[l] = 0;
    p.scratch_elems = 0;
    size_t mx = 0, gnmax = 0;
    for (int i = 0; i < c1.n_up; i++) {
        const int ch = c1.dim_upsample[i] * v1_mult(c1);              // the hidden map of a residual block is the widest one of its stage
        const size_t e = BP * ((size_t)1 << (2 * (i + 1))) * ch;
        if (e > mx) mx = e;
        const size_t gsz = groupnorm_scratch_floats(B, rows << (i + 1), cols << (i + 1), v1_hidden_groups(c1, ch));
        if (gsz > gnmax) gnmax = gsz;
    }
    v.img1 = take(p, (size_t)B * 3 * rh * rw * 4);
    v.X = take(p, mx * s); v.T1 = take(p, mx * s); v.T2 = take(p, mx * s);
    const size_t rpx = (size_t)B * rh * rw;
    v.R = take(p, rpx * v1_cpad(c1.dim_upsample[c1.n_up - 1]) * s);
    v.Y = take(p, rpx * 2 * c1.last_conv_channels * s);
    v.La = v.Lb = 0;
    if (v1_generic_out(c1)) {
        const int c4 = c1.last_conv_channels, ch4 = c4 * v1_mult(c1);
        if (c1.last_res_blocks > 0) {
            v.La = take(p, rpx * c4 * s);
            v.Lb = take(p, rpx * ch4 * s);
            const size_t gsz = groupnorm_scratch_floats(B, rh, rw, v1_hidden_groups(c1, ch4));
            if (gsz > gnmax) gnmax = gsz;
        }
    }
    v.gn = take(p, gnmax * 4);
    return v;
}

template <typename T>
static int forward_v1_impl(moge_handle* h, const void* image, int img_dtype, const PlanV1& v, float* o_points, float* o_mask, hipStream_t st) {
    const moge_v1_config& c = h->cfg1;
    const Plan& pl = v.p;
    const int B = pl.B, rh = v.rh, rw = v.rw, ph = pl.rows, pw = pl.cols;
    char* ws = h->ws + pl.base;
    float* img1 = (float*)(ws + v.img1);
    // ---- v1.py:271-278: bicubic antialiased resize to (rh, rw); a .half() model keeps this image in fp16 (values rounded on load and on store);
    // the normalisation + bilinear antialiased resize to (14 ph, 14 pw) + patchify happen in encode()'s preprocess kernel
    {
        ProfScope ps(h, st, MOGE_KC_PRE, 0, (double)B * 3 * ((double)pl.H * pl.W + (double)rh * rw) * 4);
        const int round16 = (img_dtype == 1 || img_dtype == 3) ? 1 : 0;
        if (img_dtype == 1) LCHK(launch_resize_bicubic_aa<f16>(image, img1, B, pl.H, pl.W, rh, rw, round16, st));
        else LCHK(launch_resize_bicubic_aa<float>(image, img1, B, pl.H, pl.W, rh, rw, round16, st));
    }
    CHK(encode<T>(h, img1, 0, rh, rw, pl, st));
    const double aspect = (double)rw / (double)rh;                      // Head.forward: aspect of the RESIZED image (v1.py:118)
    T* X = (T*)(ws + v.X); T* T1 = (T*)(ws + v.T1); T* T2 = (T*)(ws + v.T2);
    float* gns = (float*)(ws + v.gn);
    const T* x = (const T*)(ws + pl.feat);
    int hh = ph, ww = pw, ci = c.dim_proj;
    for (int i = 0; i < c.n_up; i++) {
        const int co = c.dim_upsample[i];
        const std::string u = S("head.upsample_blocks.%d.", i);
        {   // [x, uv] -> ConvTranspose2d(k2, s2): GEMM to 4*co columns, pixel-shuffle store; the two uv input channels are a rank-2 epilogue term
            GemmArgs g = gemm_args();
            g.a = x; g.lda = ci; g.w = P<T>(h, S("v1.up%d.wT", i)); g.ldw = ci;
            g.M = B * hh * ww; g.N = 4 * co; g.K = ci;
            g.epi = EPI_CONVT; g.bias = A(h, S("v1.up%d.biasT", i)); g.out = T1; g.Cout = co; g.pixW = ww; g.pixH = hh;
            g.uv = uv_term(A(h, S("v1.up%d.wu", i)), A(h, S("v1.up%d.wv", i)), ww, hh, aspect);
            g.uv_in = 1;
            CHK(run_gemm<T>(h, g, AMODE_LINEAR, MOGE_KC_CONV, st, ci + 2));
        }
        hh *= 2; ww *= 2;
        CHK(conv3x3<T>(h, T1, P<T>(h, S("v1.up%d.w3", i)), M(h, u + "0.1.bias"), X, B, hh, ww, co, co, 0, ACT_NONE, nullptr, nullptr, st));
        for (int j = 0; j < c.num_res_blocks; j++) {
            // ResidualConvBlock (v1.py:44-58): GN(1) -> ReLU -> 3x3 (co -> ch) -> GN(ch / 32, or 1 for "layer_norm") -> ReLU -> 3x3 (ch -> co), + x
            const std::string r = u + S("%d.layers.", 1 + j);
            const int ch = co * v1_mult(c);
            {
                ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)B * hh * ww * co * 2 * sizeof(T));
                LCHK(launch_groupnorm_relu<T>(X, T1, M(h, r + "0.weight"), M(h, r + "0.bias"), gns, B, hh, ww, co, 1, st));
            }
            CHK(conv3x3<T>(h, T1, P<T>(h, S("v1.up%d.res%d.w1", i, j)), M(h, r + "2.bias"), T2, B, hh, ww, co, ch, 0, ACT_NONE, nullptr, nullptr, st));
            {
                ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)B * hh * ww * ch * 2 * sizeof(T));
                LCHK(launch_groupnorm_relu<T>(T2, T1, M(h, r + "3.weight"), M(h, r + "3.bias"), gns, B, hh, ww, ch, v1_hidden_groups(c, ch), st));
            }
            CHK(conv3x3<T>(h, T1, P<T>(h, S("v1.up%d.res%d.w2", i, j)), M(h, r + "5.bias"), X, B, hh, ww, ch, co, 0, ACT_NONE, X, nullptr, st));
        }
        x = X; ci = co;
    }
    // ---- v1.py:127-136: bilinear resize to the resized image, uv concat, per-output [3x3 -> ReLU -> 1x1]; the two 3x3 convs run as one
    // (2 * c4 output channels), the 1x1 convs + the resize back to (H, W) + the remap run in head_final (both linear: they commute)
    const int cl = c.dim_upsample[c.n_up - 1], c4 = c.last_conv_channels, cp = v1_cpad(cl);
    T* R = (T*)(ws + v.R); T* Y = (T*)(ws + v.Y);
    {
        ProfScope ps(h, st, MOGE_KC_POST, 0, (double)B * rh * rw * cp * sizeof(T));
        const UVTerm uv = uv_term(nullptr, nullptr, rw, rh, aspect);
        LCHK(launch_resize_bilinear_uv<T>(x, R, B, hh, ww, cl, rh, rw, cp, uv.u0, uv.u1, uv.v0, uv.v1, st));
    }
    if (v1_generic_out(c)) {
        // output blocks with residual blocks and / or a 3x3 last conv (v1.py:103-109), one output at a time on dense c4-channel maps (the two
        // halves of Y): 3x3 -> [GN(1) -> ReLU -> 3x3 -> GN -> ReLU -> 3x3, + x] x n -> ReLU -> last conv; the resize back + remap stay in head_final
        const int nl = c.last_res_blocks, ks = v1_ks(c), ch4 = c4 * v1_mult(c);
        const size_t rpx = (size_t)B * rh * rw;
        T* La = (T*)(ws + v.La); T* Lb = (T*)(ws + v.Lb);
        float* gns = (float*)(ws + v.gn);
        for (int o = 0; o < 2; o++) {
            float* dst = o == 0 ? o_points : o_mask;
            if (!dst) continue;
            T* Yo = Y + (size_t)o * rpx * c4;
            CHK(conv3x3<T>(h, R, P<T>(h, "v1.out.w3") + (size_t)o * c4 * 9 * cp, A(h, "v1.out.bias") + o * c4, Yo, B, rh, rw, cp, c4, 0, nl ? ACT_NONE : ACT_RELU, nullptr, nullptr, st));
            for (int j = 0; j < nl; j++) {
                const std::string r = S("head.output_block.%d.%d.layers.", o, 1 + j);
                {
                    ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)rpx * c4 * 2 * sizeof(T));
                    LCHK(launch_groupnorm_relu<T>(Yo, La, M(h, r + "0.weight"), M(h, r + "0.bias"), gns, B, rh, rw, c4, 1, st));
                }
                CHK(conv3x3<T>(h, La, P<T>(h, S("v1.out%d.res%d.w1", o, j)), M(h, r + "2.bias"), Lb, B, rh, rw, c4, ch4, 0, ACT_NONE, nullptr, nullptr, st));
                {
                    ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)rpx * ch4 * 2 * sizeof(T));
                    LCHK(launch_groupnorm_act<T>(Lb, Lb, M(h, r + "3.weight"), M(h, r + "3.bias"), gns, B, rh, rw, ch4, v1_hidden_groups(c, ch4), MOGE_ACT_RELU, st));
                }
                CHK(conv3x3<T>(h, Lb, P<T>(h, S("v1.out%d.res%d.w2", o, j)), M(h, r + "5.bias"), Yo, B, rh, rw, ch4, c4, 0, ACT_NONE, Yo, nullptr, st));
            }
            if (nl) {
                ProfScope ps(h, st, MOGE_KC_NORM, 0, (double)rpx * c4 * 2 * sizeof(T));
                LCHK(launch_groupnorm_act<T>(Yo, Yo, nullptr, nullptr, nullptr, B, rh, rw, c4, 0, MOGE_ACT_RELU, st));       // the ReLU in front of the last conv
            }
            const std::string lk = S("head.output_block.%d.%d.", o, nl + 2);
            const int kind = o == 0 ? 0 : 3, remap = o == 0 ? c.remap_output : 0;
            if (ks == 1) {
                ProfScope ps(h, st, MOGE_KC_POST, 0, (double)rpx * c4 * sizeof(T));
                LCHK(launch_head_final<T>(kind, Yo, M(h, lk + "weight"), M(h, lk + "bias"), nullptr, nullptr, dst, B, rh, rw, c4, pl.H, pl.W, remap, st, c4));
            } else {
                // the 3x3 last conv (replicate padding) evaluated inside the resize: fp32 weights straight from the master, no intermediate map
                ProfScope ps(h, st, MOGE_KC_POST, 2.0 * B * pl.H * pl.W * 36.0 * c4 * (o == 0 ? 3 : 1), (double)rpx * c4 * sizeof(T));
                LCHK(launch_head_final_k3<T>(kind, Yo, M(h, lk + "weight"), M(h, lk + "bias"), dst, B, rh, rw, c4, pl.H, pl.W, remap, st));
            }
        }
    } else {
    CHK(conv3x3<T>(h, R, P<T>(h, "v1.out.w3"), A(h, "v1.out.bias"), Y, B, rh, rw, cp, 2 * c4, 0, ACT_RELU, nullptr, nullptr, st));
    {
        ProfScope ps(h, st, MOGE_KC_POST, 0, (double)B * rh * rw * 2 * c4 * sizeof(T));
        if (o_points)
            LCHK(launch_head_final<T>(0, Y, M(h, "head.output_block.0.2.weight"), M(h, "head.output_block.0.2.bias"), nullptr, nullptr, o_points, B, rh, rw, c4,
                                      pl.H, pl.W, c.remap_output, st, 2 * c4));
        if (o_mask)
            LCHK(launch_head_final<T>(3, Y + c4, M(h, "head.output_block.1.2.weight"), M(h, "head.output_block.1.2.bias"), nullptr, nullptr, o_mask, B, rh, rw, c4,
                                      pl.H, pl.W, 0, st, 2 * c4));
    }
    }
    h->last.valid = true; h->last.prec = TT<T>::PREC; h->last.B = B; h->last.rows = ph; h->last.cols = pw;
    h->last.bufs.clear();
    h->last.bufs["features"] = {pl.feat, {(int64_t)B * ph * pw * c.dim_proj, 1}};
    h->last.bufs["up_last"] = {v.X, {(int64_t)B * hh * ww * cl, 1}};
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" {

int moge_abi_version(void) { return MOGE_ABI_VERSION; }
const char* moge_last_error(void) { return g_err.c_str(); }

int moge_create(const moge_config* cfg, int device, moge_handle** out) {
    if (!cfg || !out) return fail(MOGE_ERR_INVALID, "null argument");
    const moge_config& c = *cfg;
    if (c.embed_dim % 128 != 0 || c.embed_dim > 1024 || c.embed_dim != c.num_heads * 64)
        return fail(MOGE_ERR_INVALID, "unsupported ViT width %d / heads %d (need head_dim 64, width %%128==0, <=1024)", c.embed_dim, c.num_heads);
    if (c.n_taps < 1 || c.n_taps > MOGE_MAX_TAPS) return fail(MOGE_ERR_INVALID, "bad n_taps");
    for (int l = 0; l < MOGE_LEVELS; l++)
        if (c.dims[l] % 8 != 0 || c.dims[l] <= 0) return fail(MOGE_ERR_INVALID, "stack dims must be positive multiples of 8");
    if (c.dims[4] > 64) return fail(MOGE_ERR_INVALID, "last level wider than 64 channels is not supported");
    if ((c.heads & MOGE_HEAD_SCALE) && (c.scale_hidden <= 0 || c.scale_hidden % 4 != 0)) return fail(MOGE_ERR_INVALID, "bad scale_hidden");
    for (int l = 0; l < MOGE_LEVELS - 1; l++)
        if (c.neck_resamplers[l] < 0 || c.neck_resamplers[l] > MOGE_RS_PIXEL_SHUFFLE || c.head_resamplers[l] < 0 || c.head_resamplers[l] > MOGE_RS_PIXEL_SHUFFLE)
            return fail(MOGE_ERR_INVALID, "bad resampler code at level %d", l);
    for (int nk = 0; nk < 2; nk++) {
        const bool neck = nk == 1;
        const int in_n = neck ? c.neck_in_norm : c.head_in_norm, hid_n = neck ? c.neck_hidden_norm : c.head_hidden_norm;
        if (in_n < 0 || in_n > MOGE_NORM_INSTANCE || hid_n < 0 || hid_n > MOGE_NORM_INSTANCE) return fail(MOGE_ERR_INVALID, "bad res-block norm code");
        if (stack_act(c, neck) < 0 || stack_act(c, neck) > MOGE_ACT_ELU) return fail(MOGE_ERR_INVALID, "bad res-block activation code");
        const int km = neck ? c.neck_hidden_mult : c.head_hidden_mult;
        if (km < 0 || km > 8) return fail(MOGE_ERR_INVALID, "dim_times_res_block_hidden must be 1 ... 8 (0 = unset), got %d", km);
        for (int l = 0; l < MOGE_LEVELS; l++) {
            // the norms run on gn_partial's / in_partial's fixed slabs (as in moge_create_v1): widths 32 ... 1024, powers of two
            const int C = c.dims[l], Ch = C * stack_mult(c, neck), nb = neck ? c.neck_res_blocks[l] : c.head_res_blocks[l];
            auto pow2 = [](int v) { return v >= 32 && v <= 1024 && (v & (v - 1)) == 0; };
            if (nb > 0 && ((in_n && !pow2(C)) || (hid_n && !pow2(Ch))))
                return fail(MOGE_ERR_INVALID, "normalised residual blocks at level %d need widths of 32 ... 1024 (powers of two), got %d (hidden %d)", l, C, Ch);
        }
    }
    HIPCHK(hipSetDevice(device));
    moge_handle* h = new moge_handle();
    h->cfg = c;
    h->device = device;
    memset(&h->prof_acc, 0, sizeof(h->prof_acc));
    build_tables(h);
    hipError_t e = hipMalloc(&h->d_status, sizeof(int));
    if (e != hipSuccess) { delete h; return fail(MOGE_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    hipMemset(h->d_status, 0, sizeof(int));
    e = hipMalloc(&h->bcast_rec, 5 * sizeof(long long));                                           // status record of moge_broadcast_weights
    if (e != hipSuccess) { hipFree(h->d_status); delete h; return fail(MOGE_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    if (hipHostMalloc((void**)&h->h_status, sizeof(int), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->h_status = nullptr; }      // (optional: moge_sync falls back to a blocking copy)
    *out = h;
    return 0;
}

int moge_create_v1(const moge_v1_config* cfg, int device, moge_handle** out) {
    if (!cfg || !out) return fail(MOGE_ERR_INVALID, "null argument");
    const moge_v1_config& c = *cfg;
    if (c.embed_dim % 128 != 0 || c.embed_dim > 1024 || c.embed_dim != c.num_heads * 64)
        return fail(MOGE_ERR_INVALID, "unsupported ViT width %d / heads %d (need head_dim 64, width %%128==0, <=1024)", c.embed_dim, c.num_heads);
    if (c.n_taps < 1 || c.n_taps > MOGE_MAX_TAPS) return fail(MOGE_ERR_INVALID, "bad n_taps");
    if (c.n_up < 1 || c.n_up > MOGE_V1_MAX_UP) return fail(MOGE_ERR_INVALID, "bad number of upsample stages");
    if (c.dim_proj % 32 != 0 || c.dim_proj <= 0) return fail(MOGE_ERR_INVALID, "dim_proj must be a positive multiple of 32");
    for (int i = 0; i < c.n_up; i++) {
        // GroupNorm(1, C) and GroupNorm(C / 32, C) run on gn_partial's fixed slabs: 256 threads must hold a whole number of pixel rows of
        // C / 8 (fp16) and C / 4 (fp32) chunks - a power of two.  Rejected HERE, not at the first forward with a generic launch error.
        const int C = c.dim_upsample[i];
        if (C != 32 && C != 64 && C != 128 && C != 256 && C != 512)
            return fail(MOGE_ERR_INVALID, "dim_upsample[%d] = %d: supported widths are 32, 64, 128, 256, 512 (GroupNorm(C / 32, C) slabs need a power of two)", i, C);
    }
    if (c.last_conv_channels != 32 && c.last_conv_channels != 64 && c.last_conv_channels != 16)
        return fail(MOGE_ERR_INVALID, "last_conv_channels must be 16, 32 or 64");
    if (c.num_res_blocks < 0 || c.num_res_blocks > 8) return fail(MOGE_ERR_INVALID, "bad num_res_blocks");
    if (c.last_res_blocks < 0 || c.last_res_blocks > 8) return fail(MOGE_ERR_INVALID, "bad last_res_blocks");
    if (c.last_conv_size != 0 && c.last_conv_size != 1 && c.last_conv_size != 3) return fail(MOGE_ERR_INVALID, "last_conv_size must be 1 or 3");
    if (c.last_res_blocks > 0) {
        const int ch4 = c.last_conv_channels * (c.hidden_mult > 0 ? c.hidden_mult : 1);
        if (c.last_conv_channels < 32 || ch4 > 1024 || (ch4 & (ch4 - 1)))
            return fail(MOGE_ERR_INVALID, "last residual blocks need last_conv_channels 32 or 64 and a power-of-two hidden width up to 1024 (got %d, hidden %d)", c.last_conv_channels, ch4);
    }
    if (c.hidden_mult < 0 || c.hidden_mult > 8) return fail(MOGE_ERR_INVALID, "dim_times_res_block_hidden must be 1 ... 8 (0 = unset), got %d", c.hidden_mult);
    if (c.res_block_norm != 0 && c.res_block_norm != MOGE_NORM_LAYER && c.res_block_norm != MOGE_NORM_GROUP) return fail(MOGE_ERR_INVALID, "res_block_norm must be group_norm or layer_norm");
    for (int i = 0; i < c.n_up && c.num_res_blocks > 0; i++) {
        const int ch = c.dim_upsample[i] * v1_mult(c);
        if (ch > 1024 || (ch & (ch - 1))) return fail(MOGE_ERR_INVALID, "dim_upsample[%d] x dim_times_res_block_hidden = %d: the hidden norm's slabs need a power of two up to 1024", i, ch);
    }
    HIPCHK(hipSetDevice(device));
    moge_handle* h = new moge_handle();
    h->version = 1;
    h->cfg1 = c;
    h->mask_thr = c.mask_threshold;
    h->bb = "backbone.";
    h->proj_fmt = "head.projects.%d";
    h->mean_key = "image_mean"; h->std_key = "image_std";
    memset(&h->cfg, 0, sizeof(h->cfg));
    h->cfg.embed_dim = c.embed_dim; h->cfg.depth = c.depth; h->cfg.num_heads = c.num_heads; h->cfg.n_taps = c.n_taps;
    for (int i = 0; i < MOGE_MAX_TAPS; i++) h->cfg.taps[i] = c.taps[i];
    h->cfg.dims[0] = c.dim_proj;
    h->cfg.heads = MOGE_HEAD_POINTS | MOGE_HEAD_MASK;
    h->cfg.remap_output = c.remap_output;
    h->device = device;
    memset(&h->prof_acc, 0, sizeof(h->prof_acc));
    build_tables(h);
    hipError_t e = hipMalloc(&h->d_status, sizeof(int));
    if (e != hipSuccess) { delete h; return fail(MOGE_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    hipMemset(h->d_status, 0, sizeof(int));
    e = hipMalloc(&h->bcast_rec, 5 * sizeof(long long));                                           // status record of moge_broadcast_weights
    if (e != hipSuccess) { hipFree(h->d_status); delete h; return fail(MOGE_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    if (hipHostMalloc((void**)&h->h_status, sizeof(int), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->h_status = nullptr; }      // (optional: moge_sync falls back to a blocking copy)
    *out = h;
    return 0;
}

void moge_destroy(moge_handle* h) {
    if (!h) return;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    if (h->master) hipFree(h->master);
    if (h->aux) hipFree(h->aux);
    for (int i = 0; i < 2; i++) if (h->packed[i]) hipFree(h->packed[i]);
    if (h->ws) hipFree(h->ws);
    if (h->u8_stage) hipFree(h->u8_stage);
    for (auto& e : h->pos_cache) hipFree(e.ptr);
    if (h->d_status) hipFree(h->d_status);
    if (h->h_status) hipHostFree(h->h_status);
    if (h->bcast_rec) hipFree(h->bcast_rec);
    for (int i = 0; i < moge_handle::MAX_SPLIT; i++) { if (h->split_st[i]) hipStreamDestroy(h->split_st[i]); if (h->ev_join[i]) hipEventDestroy(h->ev_join[i]); }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    for (int i = 0; i < moge_handle::MAX_SPLIT; i++) {
        if (h->ev_neck[i]) hipEventDestroy(h->ev_neck[i]);
        for (int k = 0; k < 3; k++) { if (h->head_st[i][k]) hipStreamDestroy(h->head_st[i][k]); if (h->ev_head[i][k]) hipEventDestroy(h->ev_head[i][k]); }
        for (int l = 0; l < MOGE_LEVELS; l++) if (h->ev_lvl[i][l]) hipEventDestroy(h->ev_lvl[i][l]);
    }
    for (auto& r : h->prof_pending) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    for (auto e : h->ev_pool) hipEventDestroy(e);
    delete h;
}

int moge_alloc_master(moge_handle* h) {
    if (!h) return fail(MOGE_ERR_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!h->master) {
        HIPCHK(hipMalloc(&h->master, h->master_floats * sizeof(float)));
        HIPCHK(hipMemset(h->master, 0, h->master_floats * sizeof(float)));
    }
    return 0;
}

int moge_master_blob(moge_handle* h, void** dev_ptr, size_t* bytes) {
    if (!h || !h->master) return fail(MOGE_ERR_NOT_LOADED, "master blob not allocated");
    if (dev_ptr) *dev_ptr = h->master;
    if (bytes) *bytes = h->master_floats * sizeof(float);
    return 0;
}

int moge_master_ready(moge_handle* h) {
    if (!h || !h->master) return fail(MOGE_ERR_NOT_LOADED, "master blob not allocated");
    HIPCHK(hipMemcpy(h->img_mean, M(h, h->mean_key), 3 * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(h->img_std, M(h, h->std_key), 3 * sizeof(float), hipMemcpyDeviceToHost));
    h->master_ready = true;
    h->aux_ready = false; h->pk_ready[0] = h->pk_ready[1] = false;
    for (auto& e : h->pos_cache) hipFree(e.ptr);
    h->pos_cache.clear();
    return 0;
}

// ---- one-time weight distribution over RCCL ------------------------------------------------------------------
// RCCL is resolved at CALL time with dlopen / dlsym - libmoge_hip.so has no link-time dependency on it (a single-GPU deployment needs no RCCL
// at all) - and from the library instance that is ALREADY loaded in the process when there is one (SONAME librccl.so.1: torch's bundled copy
// inside a PyTorch host), because the communicator handed in belongs to that instance.
namespace {
struct RcclApi {
    int (*bcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;      // ncclBroadcast(sendbuff, recvbuff, count, datatype, root, comm, stream)
    int (*allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;  // ncclAllReduce(sendbuff, recvbuff, count, datatype, op, comm, stream)
    int (*user_rank)(void*, int*) = nullptr;                                               // ncclCommUserRank
    int (*count)(void*, int*) = nullptr;                                                   // ncclCommCount
    const char* (*errstr)(int) = nullptr;                                                  // ncclGetErrorString
    bool ok = false;
};
// Only SUCCESS is cached: a host that asks before RCCL is loadable (or that loads torch's librccl later) gets another dlopen on its next call.
// Mutex-guarded (two host threads may call at once); the dlerror() text of the failing dlopen is captured at that dlopen.
const RcclApi* rccl_api(std::string& why) {
    static std::mutex mu;
    static RcclApi api;
    std::lock_guard<std::mutex> lock(mu);
    if (api.ok) return &api;
    dlerror();
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        const char* e = dlerror();
        why = e ? e : "dlopen failed";
        lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return nullptr;
    }
    RcclApi a;
    a.bcast = reinterpret_cast<decltype(a.bcast)>(dlsym(lib, "ncclBroadcast"));
    a.allreduce = reinterpret_cast<decltype(a.allreduce)>(dlsym(lib, "ncclAllReduce"));
    a.user_rank = reinterpret_cast<decltype(a.user_rank)>(dlsym(lib, "ncclCommUserRank"));
    a.count = reinterpret_cast<decltype(a.count)>(dlsym(lib, "ncclCommCount"));
    a.errstr = reinterpret_cast<decltype(a.errstr)>(dlsym(lib, "ncclGetErrorString"));
    if (!(a.bcast && a.allreduce && a.user_rank && a.count)) { why = "ncclBroadcast / ncclAllReduce / ncclCommUserRank / ncclCommCount not exported"; return nullptr; }
    a.ok = true;
    api = a;
    return &api;
}
}  // namespace

// Every rank of the communicator must call this.  A rank whose LOCAL preconditions fail (the root has no weights, bad root, allocation failure, the
// device cannot be selected) does not return early - the others would already sit in ncclBroadcast - but takes part in a 5-word status all-reduce
// first: all ranks learn that someone is not ready, or that the ranks disagree on the blob size (different model configs) or on the root, and ALL return
// an error before any payload moves.  The status record lives in the handle (allocated at create: no hipMalloc / hipFree - an implicit device-wide
// synchronisation - per call).  The only solo returns are the two cases in which no collective can be issued at all: RCCL cannot be loaded, or the
// communicator itself is broken (rank / size query fails) - then abort the communicator on the other ranks.
int moge_broadcast_weights(moge_handle* h, void* nccl_comm, int root, void* stream) {
    if (!h || !nccl_comm) return fail(MOGE_ERR_INVALID, "null argument");
    std::string why;
    const RcclApi* apip = rccl_api(why);
    if (!apip) return fail(MOGE_ERR_INVALID, "RCCL not available: librccl.so.1 could not be loaded (%s)", why.c_str());
    const RcclApi& api = *apip;
    int rank = -1, n = 0;
    int rc = api.user_rank(nccl_comm, &rank);
    if (rc == 0) rc = api.count(nccl_comm, &n);
    if (rc != 0) return fail(MOGE_ERR_INVALID, "RCCL communicator query failed: %s", api.errstr ? api.errstr(rc) : "?");
    hipStream_t st = (hipStream_t)stream;
    // ---- local checks, recorded instead of returned
    int local_status = 0;
    std::string local_msg;
    if (hipSetDevice(h->device) != hipSuccess) { local_status = MOGE_ERR_HIP; local_msg = "hipSetDevice(" + std::to_string(h->device) + ") failed"; }
    else if (root < 0 || root >= n) { local_status = MOGE_ERR_INVALID; local_msg = "root " + std::to_string(root) + " is not a rank of a " + std::to_string(n) + "-rank communicator"; }
    else if (rank == root && !h->master_ready) { local_status = MOGE_ERR_NOT_LOADED; local_msg = "the root rank has no weights to broadcast (moge_load_weights first)"; }
    else if (moge_alloc_master(h) != 0) { local_status = MOGE_ERR_HIP; local_msg = std::string("master blob allocation failed: ") + moge_last_error(); }
    // ---- agreement: min over ranks of {ok, floats, -floats, root, -root}
    long long rec[5] = {local_status == 0 ? 1 : 0, (long long)h->master_floats, -(long long)h->master_floats, root, -(long long)root};
    long long* drec = h->bcast_rec;
    const int NCCL_INT64 = 4, NCCL_MIN = 3;                     // rccl.h ncclDataType_t / ncclRedOp_t
    hipError_t he = hipMemcpyAsync(drec, rec, sizeof(rec), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) {
        rc = api.allreduce(drec, drec, 5, NCCL_INT64, NCCL_MIN, nccl_comm, st);
        if (rc == 0) he = hipMemcpyAsync(rec, drec, sizeof(rec), hipMemcpyDeviceToHost, st);
        if (rc == 0 && he == hipSuccess) he = hipStreamSynchronize(st);
    }
    if (rc != 0) return fail(MOGE_ERR_HIP, "ncclAllReduce (status agreement) failed: %s", api.errstr ? api.errstr(rc) : "?");
    HIPCHK(he);
    if (local_status != 0) return fail(local_status, "%s", local_msg.c_str());
    if (rec[0] != 1) return fail(MOGE_ERR_INVALID, "another rank of the communicator is not ready to broadcast / receive weights (its own call reports why); nothing was sent");
    if (rec[1] != -rec[2]) return fail(MOGE_ERR_INVALID, "ranks disagree on the master blob size (%lld ... %lld floats, this rank %lld): different model configs; nothing was sent",
                                       rec[1], -rec[2], (long long)h->master_floats);
    if (rec[3] != -rec[4]) return fail(MOGE_ERR_INVALID, "ranks disagree on the root rank (%lld ... %lld); nothing was sent", rec[3], -rec[4]);
    const int NCCL_FLOAT32 = 7;                                  // ncclFloat (rccl.h ncclDataType_t)
    rc = api.bcast(h->master, h->master, h->master_floats, NCCL_FLOAT32, root, nccl_comm, st);
    if (rc != 0) return fail(MOGE_ERR_HIP, "ncclBroadcast failed: %s", api.errstr ? api.errstr(rc) : "?");
    HIPCHK(hipStreamSynchronize(st));
    if (rank != root) return moge_master_ready(h);               // kernel layouts are re-packed from the received master copy on next use
    return 0;
}

int moge_load_weights(moge_handle* h, const moge_tensor_desc* descs, int n, void* stream) {
    if (!h || !descs) return fail(MOGE_ERR_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    CHK(moge_alloc_master(h));
    for (auto& kv : h->table) kv.second.loaded = false;
    for (int i = 0; i < n; i++) {
        auto it = h->table.find(descs[i].name ? descs[i].name : "");
        if (it == h->table.end()) continue;                      // strict=False
        if (it->second.numel != descs[i].numel)
            return fail(MOGE_ERR_INVALID, "tensor %s has %lld elements, config expects %lld", descs[i].name, (long long)descs[i].numel, (long long)it->second.numel);
        HIPCHK(hipMemcpyAsync(h->master + it->second.off, descs[i].data, (size_t)descs[i].numel * sizeof(float), hipMemcpyHostToDevice, st));
        it->second.loaded = true;
    }
    HIPCHK(hipStreamSynchronize(st));
    for (auto& kv : h->table)
        if (!kv.second.loaded) return fail(MOGE_ERR_MISSING_KEY, "state dict is missing %s", kv.first.c_str());
    return moge_master_ready(h);
}

int moge_set_precision(moge_handle* h, int precision, void* stream) {
    if (!h) return fail(MOGE_ERR_INVALID, "null handle");
    if (precision != MOGE_FP32 && precision != MOGE_FP16 && precision != MOGE_FP16_HALF) return fail(MOGE_ERR_INVALID, "bad precision");
    if (!h->master_ready) return fail(MOGE_ERR_NOT_LOADED, "weights not loaded");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    CHK(build_aux(h, st));
    if (precision != MOGE_FP32) CHK(pack_weights<f16>(h, st)); else CHK(pack_weights<float>(h, st));
    h->prec = precision == MOGE_FP32 ? MOGE_FP32 : MOGE_FP16;
    h->half_resid = precision == MOGE_FP16_HALF;
    return 0;
}

int moge_set_onnx_compatible_mode(moge_handle* h, int on) {
    if (!h) return fail(MOGE_ERR_INVALID, "null handle");
    h->onnx_mode = on ? 1 : 0;
    return 0;
}

int moge_workspace_bytes(moge_handle* h, int B, int H, int W, int token_rows, int token_cols, size_t* bytes) {
    if (!h || !bytes) return fail(MOGE_ERR_INVALID, "null argument");
    *bytes = forward_ws_bytes(h, make_plan(h->cfg, h->prec, B, H, W, token_rows, token_cols));
    return 0;
}

static int check_call(moge_handle* h, const void* image, int B, int H, int W, int rows, int cols, int version = 2) {
    if (!h || !image) return fail(MOGE_ERR_INVALID, "null argument");
    if (h->version != version) return fail(MOGE_ERR_INVALID, "this handle is a MoGe-%d model: use the moge_%sforward / infer entry points", h->version, h->version == 1 ? "v1_" : "");
    if (!h->master_ready) return fail(MOGE_ERR_NOT_LOADED, "weights not loaded");
    if (B <= 0 || H <= 0 || W <= 0 || rows <= 0 || cols <= 0) return fail(MOGE_ERR_INVALID, "bad shape");
    if ((long)B * rows * cols * 256 > 2000000000L) return fail(MOGE_ERR_INVALID, "batch too large for 32-bit pixel indices");
    HIPCHK(hipSetDevice(h->device));
    return 0;
}

// number of sub-batches a batch of B runs as (each on its own internal stream): BATCH_SPLIT = 0/1 off, n >= 2 -> n parts (default 2) when
// every part keeps at least BATCH_SPLIT_MIN = 3 images (batch 6 = 3 + 3: 211 -> 222 img/s; batch 4 = 2 + 2 falls into the latency-regime kernels: 215 -> 184)
static int split_parts(moge_handle* h, int B) {
    if (h->prof_on) return 1;
    int n = moge_tune_get("BATCH_SPLIT", 2);
    if (n > moge_handle::MAX_SPLIT) n = moge_handle::MAX_SPLIT;
    int min_part = moge_tune_get("BATCH_SPLIT_MIN", 3);
    if (min_part < 1) min_part = 1;                 // (a part of 0 images is not a batch)
    while (n > 1 && B / n < min_part) n--;
    return n < 2 ? 1 : n;
}
// workspace bytes a forward over plan pl needs (callers size the arena BEFORE taking pointers into it)
static size_t forward_ws_bytes(moge_handle* h, const Plan& pl) {
    const int n = split_parts(h, pl.B);
    if (n == 1) return pl.total;
    size_t off = pl.post_end;
    for (int i = 0; i < n; i++) {
        const int b0 = (int)((long)pl.B * i / n), b1 = (int)((long)pl.B * (i + 1) / n);
        off += make_plan(h->cfg, h->prec, b1 - b0, pl.H, pl.W, pl.rows, pl.cols).total;
    }
    return off > pl.total ? off : pl.total;
}

static int forward_dispatch(moge_handle* h, const void* image, int img_dtype, const Plan& pl, float* pts, float* nrm, float* mp, float* metric, hipStream_t st) {
    if (!h->pk_ready[h->prec]) CHK(moge_set_precision(h, h->half_resid ? MOGE_FP16_HALF : h->prec, st));
    const int B = pl.B;
    const int n = split_parts(h, B);
    if (n == 1) {
        CHK(ensure_ws(h, pl.total));
        if (h->prec == MOGE_FP16) return forward_impl<f16>(h, image, img_dtype, pl, pts, nrm, mp, metric, st);
        return forward_impl<float>(h, image, img_dtype, pl, pts, nrm, mp, metric, st);
    }
    // ---- n sub-batches on n internal streams ---------------------------------------------------------------------------
    for (int i = 0; i < n; i++)
        if (!h->split_st[i]) {
            HIPCHK(hipStreamCreateWithFlags(&h->split_st[i], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
        }
    if (!h->ev_fork) HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    Plan sub[moge_handle::MAX_SPLIT];
    int b0s[moge_handle::MAX_SPLIT + 1];
    size_t off = pl.post_end;                      // keep the caller-visible post buffers (mask prob, focal, ...) of the full plan
    for (int i = 0; i < n; i++) {
        b0s[i] = (int)((long)B * i / n);
        const int b1 = (int)((long)B * (i + 1) / n);
        sub[i] = make_plan(h->cfg, h->prec, b1 - b0s[i], pl.H, pl.W, pl.rows, pl.cols);
        sub[i].base = off;
        sub[i].slot = i;
        off += sub[i].total;
    }
    CHK(ensure_ws(h, off));
    const float* pos;
    CHK(get_pos(h, pl.rows, pl.cols, st, &pos));   // fill the position-embedding cache before forking
    HIPCHK(hipEventRecord(h->ev_fork, st));
    const size_t px = (size_t)pl.H * pl.W;
    const size_t img_elem = img_dtype == 1 ? 2 : 4;
    for (int i = 0; i < n; i++) {
        const size_t b0 = (size_t)b0s[i];
        HIPCHK(hipStreamWaitEvent(h->split_st[i], h->ev_fork, 0));
        const void* img_i = (const char*)image + b0 * 3 * px * img_elem;
        float* pts_i = pts ? pts + b0 * px * 3 : nullptr;
        float* nrm_i = nrm ? nrm + b0 * px * 3 : nullptr;
        float* mp_i = mp ? mp + b0 * px : nullptr;
        float* met_i = metric ? metric + b0 : nullptr;
        int rc;
        if (h->prec == MOGE_FP16) rc = forward_impl<f16>(h, img_i, img_dtype, sub[i], pts_i, nrm_i, mp_i, met_i, h->split_st[i]);
        else rc = forward_impl<float>(h, img_i, img_dtype, sub[i], pts_i, nrm_i, mp_i, met_i, h->split_st[i]);
        if (rc) return rc;
        HIPCHK(hipEventRecord(h->ev_join[i], h->split_st[i]));
    }
    for (int i = 0; i < n; i++) HIPCHK(hipStreamWaitEvent(st, h->ev_join[i], 0));
    h->last.valid = false;                          // debug taps address one contiguous batch: not available in split mode
    return 0;
}

// img_dtype 2: uint8 (B,H,W,3) as decoded from a file -> /255 in the model dtype, CHW (the reference's caller does this on the host:
// scripts/infer.py:98); the rest of the path then sees an ordinary fp32 / fp16 image.
static int ingest_image(moge_handle* h, const void*& image, int& img_dtype, int B, int H, int W, hipStream_t st) {
    if (img_dtype != 2) {
        if (img_dtype != 0 && img_dtype != 1 && img_dtype != 3)
            return fail(MOGE_ERR_INVALID, "img_dtype must be 0 (fp32 CHW), 1 (fp16 CHW), 2 (uint8 HWC) or 3 (fp32 CHW, rounded to fp16 on load)");
        return 0;
    }
    const bool half = h->prec == MOGE_FP16;
    const size_t need = (size_t)B * 3 * H * W * (half ? 2 : 4);
    if (need > h->u8_stage_bytes) {
        HIPCHK(hipStreamSynchronize(st));
        if (h->u8_stage) HIPCHK(hipFree(h->u8_stage));
        h->u8_stage = nullptr; h->u8_stage_bytes = 0;
        HIPCHK(hipMalloc(&h->u8_stage, need));
        h->u8_stage_bytes = need;
    }
    if (half) LCHK(launch_u8hwc_to_chw<f16>(image, h->u8_stage, B, H, W, st));
    else LCHK(launch_u8hwc_to_chw<float>(image, h->u8_stage, B, H, W, st));
    image = h->u8_stage;
    img_dtype = half ? 1 : 0;
    return 0;
}

int moge_forward(moge_handle* h, const void* image, int img_dtype, int B, int H, int W, int rows, int cols, const moge_outputs* out, void* stream) {
    CHK(check_call(h, image, B, H, W, rows, cols));
    if (!out) return fail(MOGE_ERR_INVALID, "null outputs");
    hipStream_t st = (hipStream_t)stream;
    CHK(ingest_image(h, image, img_dtype, B, H, W, st));
    Plan pl = make_plan(h->cfg, h->prec, B, H, W, rows, cols);
    return forward_dispatch(h, image, img_dtype, pl, out->points, out->normal, out->mask_prob, out->metric_scale, st);
}

static int post_impl(moge_handle* h, const Plan& pl, const float* pts_in, const float* nrm_in, const float* mp, const float* metric, const float* fov,
                     int flags, const moge_outputs* out, hipStream_t st) {
    const int B = pl.B, H = pl.H, W = pl.W;
    float* focal = out->focal ? out->focal : (float*)(h->ws + pl.focal);
    float* shift = out->shift ? out->shift : (float*)(h->ws + pl.shift);
    float* intr = out->intrinsics ? out->intrinsics : (float*)(h->ws + pl.intr);
    if (pts_in) {                   // (no points head, v2.py:251-281: nothing to recover - mask and masked normal only)
        ProfScope ps(h, st, MOGE_KC_RECOVER, 0, (double)B * 4096 * 16);
        LCHK(launch_recover(pts_in, mp, nullptr, fov, nullptr, B, H, W, focal, shift, intr, h->d_status, st, h->mask_thr));
    }
    {
        const size_t px = (size_t)B * H * W;
        ProfScope ps(h, st, MOGE_KC_POST, 0, (double)px * (12 + 12 + 4 + 12 + 4 + 12 + 1));
        LCHK(launch_finalize(pts_in, nrm_in, mp, metric, shift, intr, B, H, W, flags | (h->version == 1 ? 0x100 : 0), out->points, out->depth, out->normal,
                             out->mask, st, h->mask_thr));
    }
    return 0;
}

int moge_infer(moge_handle* h, const void* image, int img_dtype, int B, int H, int W, int rows, int cols, const float* fov_x_deg, int flags,
               const moge_outputs* out, void* stream) {
    CHK(check_call(h, image, B, H, W, rows, cols));
    if (!out) return fail(MOGE_ERR_INVALID, "null outputs");
    const moge_config& c = h->cfg;
    // every head is optional (v2.py:46-56): without a points head infer() returns the mask (no `depth > 0` term) and the masked normal (v2.py:251-298)
    const bool has_pts = (c.heads & MOGE_HEAD_POINTS) != 0;
    if (has_pts && (!out->points || !out->depth)) return fail(MOGE_ERR_INVALID, "points and depth output buffers are required");
    if (!has_pts && !(c.heads & (MOGE_HEAD_MASK | MOGE_HEAD_NORMAL))) return fail(MOGE_ERR_INVALID, "the model has no points, mask or normal head: infer() has nothing to return");
    hipStream_t st = (hipStream_t)stream;
    CHK(ingest_image(h, image, img_dtype, B, H, W, st));
    Plan pl = make_plan(c, h->prec, B, H, W, rows, cols);
    CHK(ensure_ws(h, forward_ws_bytes(h, pl)));
    float* mp = (c.heads & MOGE_HEAD_MASK) ? (out->mask_prob ? out->mask_prob : (float*)(h->ws + pl.maskprob)) : nullptr;
    float* nrm = (c.heads & MOGE_HEAD_NORMAL) ? out->normal : nullptr;
    float* metric = (c.heads & MOGE_HEAD_SCALE) ? (out->metric_scale ? out->metric_scale : (float*)(h->ws + pl.metric)) : nullptr;
    CHK(forward_dispatch(h, image, img_dtype, pl, has_pts ? out->points : nullptr, nrm, mp, metric, st));
    return post_impl(h, pl, has_pts ? out->points : nullptr, nrm, mp, metric, fov_x_deg, flags, out, st);
}

static int v1_check(moge_handle* h, const void* image, int B, int H, int W, int rh, int rw, void* stream) {
    CHK(check_call(h, image, B, H, W, rh / 14, rw / 14, 1));
    if (rh < 14 || rw < 14) return fail(MOGE_ERR_INVALID, "resized image %dx%d is smaller than one 14x14 patch", rh, rw);
    // lazy weight packing runs on the CALL's stream (as forward_dispatch does for v2): on the NULL stream it would race the forward of a
    // caller that uses a hipStreamNonBlocking stream and never called moge_set_precision
    if (!h->pk_ready[h->prec]) CHK(moge_set_precision(h, h->half_resid ? MOGE_FP16_HALF : h->prec, stream));
    return 0;
}

int moge_v1_forward(moge_handle* h, const void* image, int img_dtype, int B, int H, int W, int resized_h, int resized_w, const moge_outputs* out, void* stream) {
    CHK(v1_check(h, image, B, H, W, resized_h, resized_w, stream));
    if (!out) return fail(MOGE_ERR_INVALID, "null outputs");
    hipStream_t st = (hipStream_t)stream;
    CHK(ingest_image(h, image, img_dtype, B, H, W, st));
    const PlanV1 v = make_plan_v1(h, h->prec, B, H, W, resized_h, resized_w);
    CHK(ensure_ws(h, v.p.total));
    if (h->prec == MOGE_FP16) return forward_v1_impl<f16>(h, image, img_dtype, v, out->points, out->mask_prob, st);
    return forward_v1_impl<float>(h, image, img_dtype, v, out->points, out->mask_prob, st);
}

int moge_v1_infer(moge_handle* h, const void* image, int img_dtype, int B, int H, int W, int resized_h, int resized_w, const float* fov_x_deg, int flags,
                  const moge_outputs* out, void* stream) {
    CHK(v1_check(h, image, B, H, W, resized_h, resized_w, stream));
    if (!out || !out->points || !out->depth) return fail(MOGE_ERR_INVALID, "points and depth output buffers are required");
    hipStream_t st = (hipStream_t)stream;
    CHK(ingest_image(h, image, img_dtype, B, H, W, st));
    const PlanV1 v = make_plan_v1(h, h->prec, B, H, W, resized_h, resized_w);
    CHK(ensure_ws(h, v.p.total));
    float* mp = out->mask_prob ? out->mask_prob : (float*)(h->ws + v.p.maskprob);
    int rc = h->prec == MOGE_FP16 ? forward_v1_impl<f16>(h, image, img_dtype, v, out->points, mp, st)
                                  : forward_v1_impl<float>(h, image, img_dtype, v, out->points, mp, st);
    if (rc) return rc;
    return post_impl(h, v.p, out->points, nullptr, mp, nullptr, fov_x_deg, flags, out, st);
}

int moge_postprocess(moge_handle* h, const float* points_in, const float* normal_in, const float* mask_prob_in, const float* metric_scale_in,
                     int B, int H, int W, const float* fov_x_deg, int flags, const moge_outputs* out, void* stream) {
    if (!h || !points_in || !out || !out->points || !out->depth) return fail(MOGE_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    Plan pl;
    pl.B = B; pl.H = H; pl.W = W;
    pl.focal = take(pl, (size_t)B * 4); pl.shift = take(pl, (size_t)B * 4); pl.intr = take(pl, (size_t)B * 36);
    CHK(ensure_ws(h, pl.total));
    return post_impl(h, pl, points_in, normal_in, mask_prob_in, metric_scale_in, fov_x_deg, flags, out, st);
}

int moge_depth_edge_mask(moge_handle* h, const float* depth, const unsigned char* mask, int B, int H, int W, float rtol, unsigned char* out, void* stream) {
    if (!h || !depth || !out || B < 1 || H < 1 || W < 1) return fail(MOGE_ERR_INVALID, "null / empty argument");
    HIPCHK(hipSetDevice(h->device));
    LCHK(launch_depth_edge_mask(depth, mask, out, B, H, W, rtol, (hipStream_t)stream));
    return 0;
}

int moge_cast_f16(const float* src, void* dst_f16, int64_t n, void* stream) {
    if (!src || !dst_f16 || n < 0) return fail(MOGE_ERR_INVALID, "moge_cast_f16: null argument or negative count");
    if (n == 0) return 0;
    LCHK((launch_convert<float, f16>(src, dst_f16, (long)n, (hipStream_t)stream)));
    return 0;
}

int moge_sync(moge_handle* h, void* stream) {
    if (!h) return fail(MOGE_ERR_INVALID, "null handle");
    // ONE host wait: the status word is copied to pinned host memory on the stream, behind the work it reports on.  (Rounds 1-5 waited for the stream and then
    // ran a blocking 4-byte copy: two host round trips of ~90 us each per call - 1.4 % of a single-image infer().)
    int stv = 0;
    if (h->h_status) {
        HIPCHK(hipMemcpyAsync(h->h_status, h->d_status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));      // (polling the pinned word instead of this blocking wait measured level: profiles/r06p_ab_SYNC_SPIN_US_b1.log)
        stv = *(volatile int*)h->h_status;
    } else {
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        HIPCHK(hipMemcpy(&stv, h->d_status, sizeof(int), hipMemcpyDeviceToHost));
    }
    if (stv != 0) {
        HIPCHK(hipMemset(h->d_status, 0, sizeof(int)));
        return fail(stv, "Residuals are not finite in the initial point.");
    }
    return 0;
}

int moge_profile_enable(moge_handle* h, int on) {
    if (!h) return fail(MOGE_ERR_INVALID, "null handle");
    h->prof_on = on != 0;
    return 0;
}

int moge_profile_read(moge_handle* h, moge_profile* out, int reset) {
    if (!h || !out) return fail(MOGE_ERR_INVALID, "null argument");
    for (auto& r : h->prof_pending) {
        HIPCHK(hipEventSynchronize(r.e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, r.e0, r.e1));
        h->prof_acc.ms[r.cls] += ms;
        h->prof_acc.flops[r.cls] += r.flops;
        h->prof_acc.bytes[r.cls] += r.bytes;
        h->prof_acc.launches[r.cls] += 1;
        h->ev_pool.push_back(r.e0);
        h->ev_pool.push_back(r.e1);
    }
    h->prof_pending.clear();
    *out = h->prof_acc;
    if (reset) memset(&h->prof_acc, 0, sizeof(h->prof_acc));
    return 0;
}

int moge_debug_tap(moge_handle* h, const char* name, float* dst, int64_t cap, int64_t* numel, void* stream) {
    if (!h || !name) return fail(MOGE_ERR_INVALID, "null argument");
    if (!h->last.valid) return fail(MOGE_ERR_INVALID, "no forward has run");
    auto it = h->last.bufs.find(name);
    if (it == h->last.bufs.end()) return fail(MOGE_ERR_INVALID, "unknown tap %s", name);
    const int64_t n = it->second.second.first;
    if (numel) *numel = n;
    if (!dst) return 0;
    if (cap < n) return fail(MOGE_ERR_INVALID, "tap buffer too small");
    hipStream_t st = (hipStream_t)stream;
    const char* src = h->ws + it->second.first;
    if (it->second.second.second == 0 || h->last.prec == MOGE_FP32) LCHK((launch_convert<float, float>(src, dst, n, st)));
    else LCHK((launch_convert<f16, float>(src, dst, n, st)));
    return 0;
}

}  // extern "C"
