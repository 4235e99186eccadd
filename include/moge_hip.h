/*
 * moge_hip.h - C ABI of libmoge_hip.so: the MI355X-native (gfx950) implementation of the
 * microsoft/MoGe `moge.model.v2.MoGeModel.infer()` hot path.
 *
 * Plain C, plain pointers and sizes, no torch types.  Every entry point names the reference interface it
 * replaces (paths relative to the reference checkout).  The Python host (moge_amd/model/v2.py) binds these
 * with ctypes and mirrors the reference's MoGeModel surface on top; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - all functions return 0 on success, a negative moge_status otherwise; moge_last_error() gives the text
 *   - device pointers are raw HIP device addresses in the calling process (e.g. torch.Tensor.data_ptr())
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on it
 *   - a handle is bound to one device and is not thread-safe; different handles are independent
 */
#ifndef MOGE_HIP_H
#define MOGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOGE_ABI_VERSION 5
#define MOGE_MAX_TAPS 8
#define MOGE_LEVELS 5

typedef enum moge_status {
    MOGE_OK = 0,
    MOGE_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    MOGE_ERR_HIP = -2,          /* a HIP runtime call failed */
    MOGE_ERR_NOT_LOADED = -3,   /* weights missing */
    MOGE_ERR_MISSING_KEY = -4,  /* a state-dict tensor the config requires was not supplied */
    MOGE_ERR_NONFINITE = -5     /* focal/shift solve saw non-finite residuals at x0 (scipy raises ValueError) */
} moge_status;

typedef enum moge_precision {
    MOGE_FP32 = 0,        /* fp32 storage, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the parity mode */
    MOGE_FP16 = 1,        /* fp16 storage, fp32 accumulate (v_mfma_f32_16x16x32_f16), fp32 residual stream: fp32 weights + use_fp16=True, i.e. the
                           * reference under torch.autocast (v2.py:241: LayerNorm / residual adds stay fp32 there) */
    MOGE_FP16_HALF = 2    /* the same kernels and packed weights with the residual stream itself in fp16: `model.half()` (scripts/infer.py:83-84,
                           * block.py:110-112 on half tensors) - 4 instead of 10 bytes per element in the proj / fc2 epilogues */
} moge_precision;

typedef enum moge_remap { MOGE_REMAP_LINEAR = 0, MOGE_REMAP_SINH = 1, MOGE_REMAP_EXP = 2, MOGE_REMAP_SINH_EXP = 3 } moge_remap;

/* heads-present bitmask */
#define MOGE_HEAD_POINTS 1
#define MOGE_HEAD_NORMAL 2
#define MOGE_HEAD_MASK   4
#define MOGE_HEAD_SCALE  8

/* ConvStack options (moge/model/modules.py:18-67, 139-181, 195-240).  Resampler of level l -> l + 1 (x2 up-samplers only: the decoder stacks): */
typedef enum moge_resampler { MOGE_RS_CONV_TRANSPOSE = 0, MOGE_RS_BILINEAR = 1, MOGE_RS_NEAREST = 2, MOGE_RS_PIXEL_SHUFFLE = 3 } moge_resampler;
/* in_norm / hidden_norm of the residual blocks (modules.py:47-58): Identity, GroupNorm(1, C) ("layer_norm"), GroupNorm(C / 32, C) ("group_norm"),
 * InstanceNorm2d(C) ("instance_norm": per-channel statistics, no affine parameters) */
typedef enum moge_res_norm { MOGE_NORM_NONE = 0, MOGE_NORM_LAYER = 1, MOGE_NORM_GROUP = 2, MOGE_NORM_INSTANCE = 3 } moge_res_norm;
/* activation of the residual blocks (modules.py:31-40): ReLU, LeakyReLU(0.2), SiLU, ELU(1) */
typedef enum moge_activation { MOGE_ACT_RELU = 0, MOGE_ACT_LEAKY_RELU = 1, MOGE_ACT_SILU = 2, MOGE_ACT_ELU = 3 } moge_activation;

/* Mirrors the checkpoint's `model_config` (moge/model/v2.py:29-56, configs/train/v2.json:238-285): 5 levels, replicate padding.
 * The released layout - resamplers [conv_transpose x3, bilinear], res-block norms "none", ReLU, hidden width = width - runs on the fused
 * throughput kernels; the other resamplers / norms (ABI v3) and the other activations / instance_norm / dim_times_res_block_hidden > 1
 * (ABI v4) run on the generic kernels of the same library.  Not representable: the x0.5 resamplers (pixel_unshuffle, avg_pool, max_pool) -
 * MoGeModel.forward hands level l a map of 2^l x the token grid (v2.py:154-160), so ConvStack.forward's `x + feature` (modules.py:247-249)
 * is a shape error in the reference itself for any of them. */
typedef struct moge_config {
    int32_t embed_dim;                 /* ViT width D (384 / 768 / 1024)            vision_transformer.py:351-390 */
    int32_t depth;                     /* ViT blocks                                                               */
    int32_t num_heads;                 /* head_dim must be 64                                                      */
    int32_t n_taps;                    /* len(intermediate_layers)                  modules.py:82                  */
    int32_t taps[MOGE_MAX_TAPS];       /* block indices whose output is tapped                                     */
    int32_t dims[MOGE_LEVELS];         /* dim_res_blocks (dims[0] == encoder dim_out)                              */
    int32_t neck_res_blocks[MOGE_LEVELS];
    int32_t head_res_blocks[MOGE_LEVELS];
    int32_t heads;                     /* MOGE_HEAD_* bitmask                                                      */
    int32_t scale_hidden;              /* scale_head dims = [D, scale_hidden, scale_hidden, 1]                     */
    int32_t remap_output;              /* moge_remap                                v2.py:122-136                  */
    int32_t neck_resamplers[MOGE_LEVELS - 1];   /* moge_resampler per level transition       modules.py:139-181            */
    int32_t head_resamplers[MOGE_LEVELS - 1];   /* (all heads share one layout)                                            */
    int32_t neck_in_norm, neck_hidden_norm;     /* moge_res_norm                             modules.py:47-60              */
    int32_t head_in_norm, head_hidden_norm;
    int32_t neck_activation, head_activation;   /* moge_activation, 0 = ReLU                 modules.py:31-40, 203         */
    int32_t neck_hidden_mult, head_hidden_mult; /* dim_times_res_block_hidden (0 reads as 1) modules.py:199, 222           */
} moge_config;

/* Mirrors the `model_config` of a MoGe-1 checkpoint (moge/model/v1.py:148-163): group_norm / layer_norm residual blocks,
 * dim_times_res_block_hidden 1 ... 8 (configs/train/v1.json:31 trains with 2), last_res_blocks 0 ... 8, last_conv_size 1 or 3, head outputs
 * [3 (points), 1 (mask)].  The default output block (no last residual blocks, 1x1 last conv: also what configs/train/v1.json uses) runs fused;
 * the other layouts run per output on the generic kernels. */
#define MOGE_V1_MAX_UP 4
typedef struct moge_v1_config {
    int32_t embed_dim, depth, num_heads;      /* ViT (head_dim 64)                                                          */
    int32_t n_taps;                           /* number of tapped blocks                                                    */
    int32_t taps[MOGE_MAX_TAPS];              /* their indices (an int `intermediate_layers` n = the LAST n blocks)         */
    int32_t dim_proj;                         /* Head.projects output channels                          v1.py:79-81         */
    int32_t n_up;                             /* len(dim_upsample)                                                          */
    int32_t dim_upsample[MOGE_V1_MAX_UP];     /* channels after each [ConvTranspose2d k2 s2, 3x3, res blocks] stage         */
    int32_t num_res_blocks;                   /* ResidualConvBlocks per stage                           v1.py:86            */
    int32_t last_conv_channels;               /* hidden width of the output blocks                      v1.py:105-110       */
    int32_t remap_output;                     /* moge_remap                                             v1.py:253-267       */
    float mask_threshold;                     /* validity = raw mask output > mask_threshold            v1.py:358           */
    int32_t hidden_mult;                      /* dim_times_res_block_hidden (0 reads as 1)              v1.py:69, 85        */
    int32_t res_block_norm;                   /* hidden norm: MOGE_NORM_GROUP (or 0) = GroupNorm(Ch / 32, Ch), MOGE_NORM_LAYER = GroupNorm(1, Ch)   v1.py:47 */
    int32_t last_res_blocks;                  /* ResidualConvBlocks inside each output block            v1.py:106           */
    int32_t last_conv_size;                   /* kernel of the last conv, 1 or 3 (0 reads as 1)         v1.py:108           */
} moge_v1_config;

/* One state-dict entry handed to moge_load_weights: fp32, contiguous, host memory. */
typedef struct moge_tensor_desc {
    const char* name;                  /* reference state-dict key, e.g. "encoder.backbone.blocks.0.attn.qkv.weight" */
    const float* data;
    int64_t numel;
} moge_tensor_desc;

/* Device output buffers of one call, owned by the caller (NULL = not wanted / head absent). */
typedef struct moge_outputs {
    float* points;        /* (B,H,W,3) */
    float* depth;         /* (B,H,W)   infer only */
    float* normal;        /* (B,H,W,3) */
    float* mask_prob;     /* (B,H,W)   forward: sigmoid probability */
    uint8_t* mask;        /* (B,H,W)   infer: validity mask, 0/1 */
    float* intrinsics;    /* (B,3,3)   infer only */
    float* metric_scale;  /* (B,)      */
    float* focal;         /* (B,) optional: recovered focal (relative to half diagonal) */
    float* shift;         /* (B,) optional: recovered z shift */
} moge_outputs;

/* infer flags (v2.py:199-201) */
#define MOGE_FORCE_PROJECTION 1
#define MOGE_APPLY_MASK 2

typedef struct moge_handle moge_handle;

/* Kernel classes for the built-in HIP-event profiler (bench.py roofline). */
/* MOGE_KC_GEMM_PP: launches of the ViT / out-projection linear layers that ran on the persistent ping-pong throughput kernel (gemm_pp128p_kernel; K < 192: its one-tile form gemm_pp128m16_kernel) -
 * the roofline object of bench.py; MOGE_KC_GEMM: the same layers on the latency-regime kernels + the patch-embed GEMM. */
enum { MOGE_KC_GEMM = 0, MOGE_KC_ATTN = 1, MOGE_KC_CONV = 2, MOGE_KC_NORM = 3, MOGE_KC_PRE = 4, MOGE_KC_POST = 5,
       MOGE_KC_RECOVER = 6, MOGE_KC_GEMM_PP = 7, MOGE_KC_COUNT = 8 };
typedef struct moge_profile {
    double ms[MOGE_KC_COUNT];        /* summed kernel time per class (HIP events on the launch stream) */
    double flops[MOGE_KC_COUNT];     /* algorithmic FLOPs (2*MAC) launched per class */
    double bytes[MOGE_KC_COUNT];     /* algorithmic HBM bytes (compulsory reads+writes) per class */
    int64_t launches[MOGE_KC_COUNT];
} moge_profile;

int moge_abi_version(void);
const char* moge_last_error(void);

/* replaces MoGeModel.__init__ (v2.py:30-57): build the model skeleton for `cfg` on HIP device `device`. */
int moge_create(const moge_config* cfg, int device, moge_handle** out);
void moge_destroy(moge_handle* h);

/* replaces moge.model.v1.MoGeModel.__init__ (v1.py:148-205).  The returned handle takes the same moge_load_weights / master-blob /
 * moge_set_precision / moge_sync / profiler calls as a MoGe-2 handle (state-dict keys: "backbone.*", "head.*", "image_mean", "image_std"). */
int moge_create_v1(const moge_v1_config* cfg, int device, moge_handle** out);

/* replaces nn.Module.load_state_dict (v2.py:105): upload the fp32 master copy of every tensor the config
 * needs.  Unknown names are ignored (strict=False); a missing required tensor -> MOGE_ERR_MISSING_KEY. */
int moge_load_weights(moge_handle* h, const moge_tensor_desc* descs, int n, void* stream);

/* Multi-GPU weight distribution: the fp32 master blob is one contiguous device buffer whose
 * layout depends on the config only.  Rank 0 loads it with moge_load_weights; other ranks call
 * moge_alloc_master, receive the bytes with an RCCL broadcast into the returned pointer (the host does that
 * with torch.distributed), then call moge_master_ready. */
int moge_alloc_master(moge_handle* h);
int moge_master_blob(moge_handle* h, void** dev_ptr, size_t* bytes);
int moge_master_ready(moge_handle* h);
/* The same distribution for a host WITHOUT torch: `nccl_comm` is an ncclComm_t of the calling process (one rank per GPU,
 * created by the host with ncclCommInitRank), passed as void* so that this header needs no RCCL include.  Rank `root` must hold loaded
 * weights; every rank calls this once: ncclBroadcast of the fp32 master blob over xGMI on `stream`, then (non-root ranks) moge_master_ready.
 * RCCL is bound at call time (dlopen of the librccl.so.1 already in the process, else the system one): no link-time dependency.
 * Failure behaviour: before any payload moves, the ranks all-reduce a status record (ready flag, blob size, root); if ANY rank is not ready
 * (root without weights, allocation failure) or the ranks disagree on the blob size (different configs) or on the root, EVERY rank returns
 * an error and nothing is sent - no rank is left blocked inside the collective.  Only a rank that cannot reach RCCL at all (or holds a broken
 * communicator) returns alone; the host must then abort the communicator on the others.
 * The reference has no counterpart - it is a single-process PyTorch module (moge/model/v2.py:76-107 loads one checkpoint per process). */
int moge_broadcast_weights(moge_handle* h, void* nccl_comm, int root, void* stream);

/* replaces nn.Module.half()/.float() (scripts/infer.py:82-84) and the autocast switch of infer(use_fp16=...) (v2.py:241): select the compute
 * precision - MOGE_FP32 (.float(), use_fp16=False), MOGE_FP16 (fp32 weights + use_fp16=True: the reference runs torch.autocast, residual stream
 * fp32) or MOGE_FP16_HALF (.half(): every tensor fp16, the residual stream included).  Packs the kernel-layout weight set of that storage type on
 * first use (the fp32 and the fp16 set may both stay resident; the two fp16 modes share one). */
int moge_set_precision(moge_handle* h, int precision, void* stream);

/* replaces the `MoGeModel.onnx_compatible_mode` setter (v2.py:67-74, docs/onnx.md): the forward the reference exports to ONNX - the 14x
 * image resize without antialiasing (modules.py:121) and the position embedding resampled by output size instead of the scale-factor
 * kludge, never bypassed (vision_transformer.py:192,202-210).  Affects forward and infer of this handle until switched off. */
int moge_set_onnx_compatible_mode(moge_handle* h, int on);

/* bytes of device workspace a call with these shapes needs (grown lazily by forward/infer). */
int moge_workspace_bytes(moge_handle* h, int B, int H, int W, int token_rows, int token_cols, size_t* bytes);

/* replaces MoGeModel.forward (v2.py:138-192).  image: device, (B,3,H,W), fp32 (img_dtype 0) or fp16 (1),
 * values in [0,1]; or img_dtype 2: uint8 (B,H,W,3) as decoded from a file - the library then does the caller's
 * `image / 255` + HWC->CHW + cast to the model dtype (scripts/infer.py:98, v2.py:229) on the device (4x / 2x less
 * PCIe traffic than uploading floats); or img_dtype 3: fp32 (B,3,H,W) whose values are rounded to fp16 as they are read - the
 * `image.to(dtype=self.dtype)` of a .half() model (v2.py:229) without a separate cast pass.  token_rows/cols = base_h/base_w computed by the host exactly as v2.py:142-147.
 * Writes points (remapped), normal (unit), mask_prob, metric_scale. */
int moge_forward(moge_handle* h, const void* image, int img_dtype, int B, int H, int W, int token_rows, int token_cols,
                 const moge_outputs* out, void* stream);

/* replaces MoGeModel.infer (v2.py:194-303) after the host has resolved num_tokens: forward + focal/shift
 * recovery (geometry_torch.py:115-170; MINPACK lmdif in fp64 on device) + intrinsics + re-projection +
 * metric scale + masking.  fov_x_deg: NULL, or device pointer to B floats (degrees).
 * Every head is optional (v2.py:46-56): a model without a points head returns the validity mask (probability > 0.5, no `depth > 0` term)
 * and the masked normal only (v2.py:251-298) - points / depth / intrinsics buffers are then ignored; without a mask head nothing is masked;
 * without a scale head nothing is scaled. */
int moge_infer(moge_handle* h, const void* image, int img_dtype, int B, int H, int W, int token_rows, int token_cols,
               const float* fov_x_deg, int flags, const moge_outputs* out, void* stream);

/* replace moge.model.v1.MoGeModel.forward (v1.py:269-300) and .infer (v1.py:302-391) on a handle made by moge_create_v1.  The host passes
 * (resized_h, resized_w) = the size v1.py:272-274 computes from num_tokens (Python float arithmetic + int() truncation), the library does the
 * bicubic antialiased resize, normalisation, bilinear antialiased resize to multiples of 14, the ViT, the head, the resize back and the
 * remap.  forward writes points (B,H,W,3) and mask_prob (B,H,W) = the RAW mask output (no activation in MoGe-1); infer additionally writes
 * depth, the validity mask (raw > mask_threshold; no depth > 0 term in v1), intrinsics.  normal / metric_scale do not exist in MoGe-1. */
int moge_v1_forward(moge_handle* h, const void* image, int img_dtype, int B, int H, int W, int resized_h, int resized_w,
                    const moge_outputs* out, void* stream);
int moge_v1_infer(moge_handle* h, const void* image, int img_dtype, int B, int H, int W, int resized_h, int resized_w,
                  const float* fov_x_deg, int flags, const moge_outputs* out, void* stream);

/* replaces the post-processing half of infer (v2.py:246-289) on caller-supplied forward outputs: used to
 * test the recovery path in isolation.  points/normal/mask_prob are read; all outputs written. */
int moge_postprocess(moge_handle* h, const float* points_in, const float* normal_in, const float* mask_prob_in,
                     const float* metric_scale_in, int B, int H, int W, const float* fov_x_deg, int flags,
                     const moge_outputs* out, void* stream);

/* replaces the caller-side mesh clean-up `mask & ~utils3d.np.depth_map_edge(depth, rtol=threshold)` (scripts/infer.py:127;
 * utils3d is an un-vendored dependency - algorithm restated in csrc/post.hip and oracle/caller_side.py, parity unpinned):
 * depth (B,H,W) fp32 with +inf outside the mask, mask (B,H,W) bytes or NULL, out (B,H,W) bytes = mask && !edge. */
int moge_depth_edge_mask(moge_handle* h, const float* depth, const unsigned char* mask, int B, int H, int W, float rtol,
                         unsigned char* out, void* stream);

/* replaces the per-key `.half()` of MoGeModel.forward on a half model (moge/model/v2.py:386-387): n fp32 values -> fp16
 * (round to nearest even, as torch), so that forward() too returns without a torch op between the kernels and the caller. */
int moge_cast_f16(const float* src, void* dst_f16, int64_t n, void* stream);      /* stateless: runs on the current device */

/* Synchronise `stream` and report the sticky device-side status of the calls since the last sync
 * (MOGE_ERR_NONFINITE if a recovery solve saw non-finite residuals). */
int moge_sync(moge_handle* h, void* stream);

/* HIP-event profiler: when enabled every kernel launch is bracketed by events on its stream. */
int moge_profile_enable(moge_handle* h, int on);
int moge_profile_read(moge_handle* h, moge_profile* out, int reset);   /* synchronises pending events */

/* Debug taps for stage-level parity: copy an internal activation of the LAST forward to a caller buffer as
 * fp32.  name: "tokens0", "tap<k>", "cls", "features", "neck<l>", "head_<points|normal|mask>_x4".
 * Layout is the library's own (token-major / NHWC); *numel receives the element count. */
int moge_debug_tap(moge_handle* h, const char* name, float* dst, int64_t dst_capacity, int64_t* numel, void* stream);

/* Runtime tuning / A-B switches (same keys as the MOGE_<KEY> environment variables; tests and tools only):
 *   GEMM_PP, PP_ROW128, ATTN_PP, ATTN_NW, BATCH_SPLIT, GLDS_VARIANT, ...   The library's defaults are the tuned ones. */
void moge_tune_set(const char* key, int value);

/* ---- per-kernel test entry points (stage-level parity; tests/ only) -------------------------------- */
/* C[M,N] = A[M,K] * W[N,K]^T + bias, fp32 in/out on device; computed in `precision`. act: 0 none 1 relu 2 gelu */
int moge_test_gemm(int precision, const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                   int act, void* stream);
/* The same GEMM through every fused epilogue of the ViT / decoder linear layers (gemm.hip + gemm_pp.hip), so that the production fp16
 * throughput kernel (gemm_pp128p_kernel, persistent; gemm_pp128m16_kernel for K < 192: selected when N % 256 == 0, K % 64 == 0 and the problem has >= PP_MIN_TILES 256x256 tiles, or forced
 * with moge_tune_set("PP_MIN_TILES", 0)) and the latency-regime kernels (moge_tune_set("GEMM_PP", 0)) can each be compared with a plain
 * fp32 reference and with each other.  All pointers are DEVICE fp32 unless noted; acc = A W^T.
 *   MOGE_TG_STORE  out[m][n]  = act(lnfold(acc) + bias[n] (+ wu[n] u(x) + wv[n] v(y)))            attention.py:72, mlp.py:35, modules.py:128-131
 *   MOGE_TG_RESID  xres[m][n] += gamma[n] (acc + bias[n]); optional x16_out (fp16 copy, returned as fp32) and ln_part_out[m][N/32][2]
 *                  = (sum, sum of squares) of every 32-column group of the updated row               block.py:111-112, layer_scale.py:27
 *                  xres == NULL: the fp16 residual stream of a `.half()` model - x16_out is IN / OUT (fp32 values, rounded to fp16 on the
 *                  way in): x16 <- fp16(x16 + gamma (acc + bias)); ln_part_out (optional) = the statistics of the ROUNDED row
 *   MOGE_TG_QKV    q/k/v_out (B,nh,Ntok,64) = head-major split of lnfold(acc) + bias, q scaled by qscale          attention.py:72-74
 *   MOGE_TG_CONVT  out (B,2 pixH,2 pixW,Cout): n = (dy*2+dx)*Cout + co of pixel m = (b*pixH + y)*pixW + x goes to (2y+dy, 2x+dx)   modules.py:162
 * lnfold(acc) = ln_mr[m][1] * (acc - ln_mr[m][0] * ln_c[n]) when ln_mr != NULL (LayerNorm folded into the consumer GEMM), else acc.
 * u(x) = linspace(u0,u1,pixW)[m % pixW], v(y) = linspace(v0,v1,pixH)[(m / pixW) % pixH] when wu != NULL. */
enum { MOGE_TG_STORE = 0, MOGE_TG_RESID = 1, MOGE_TG_QKV = 2, MOGE_TG_CONVT = 3 };
typedef struct moge_test_gemm_args {
    int32_t precision, kind, act;            /* act: 0 none 1 relu 2 gelu (STORE only) */
    int32_t M, N, K;
    const float* A; const float* W; const float* bias;
    float* out;                              /* STORE: [M][N]; CONVT: (B,2pixH,2pixW,Cout) */
    const float* ln_mr; const float* ln_c;   /* [M][2] (mean, rstd), [N] */
    const float* wu; const float* wv; float u0, u1, v0, v1;
    int32_t pixW, pixH, Cout;
    float* xres; const float* gamma; float* x16_out; float* ln_part_out;      /* RESID */
    float* q_out; float* k_out; float* v_out; int32_t nh, Ntok; float qscale; /* QKV: M = B*Ntok, N = 3*nh*64 */
} moge_test_gemm_args;
int moge_test_gemm_ex(const moge_test_gemm_args* args, void* stream);
/* LayerNorm rows of x[rows, D], eps 1e-6 */
int moge_test_layernorm(int precision, const float* x, const float* w, const float* b, float* y, int rows, int D, void* stream);
/* softmax(q k^T / 8) v for q,k,v (B,nh,N,64) fp32 -> o (B,N,nh*64) */
int moge_test_attention(int precision, const float* q, const float* k, const float* v, float* o, int B, int nh, int N, void* stream);
/* 3x3 replicate-padded conv, NHWC: x (B,H,W,Cin), w (Cout,Cin,3,3) torch layout, y (B,H,W,Cout); relu_in applies ReLU to x */
int moge_test_conv3x3(int precision, const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                      int Cin, int Cout, int relu_in, void* stream);
/* The same conv with the pieces the decoder fuses into it (conv_pp.hip flavours), each against F.conv2d / F.pixel_shuffle in tests/:
 *   y = [add +] act(conv3x3(relu_in ? relu(x) : x) + bias [+ side_w . side] [+ wu u(x) + wv v(y)])          modules.py:53-66, 148-181, 245
 *   up2: bilinear x2 + 3x3 as the 4-phase conv + pixel shuffle, y (B,2H,2W,Cout), uv at the OUTPUT resolution    modules.py:155-159
 *   w2 != NULL: the fused residual block  y = x + conv2(relu(conv1(relu(x)) + bias)) + bias2  in ONE launch       modules.py:47-68
 *   dot_w != NULL (with up2): y[..., e] = sum_c dot_w[e][c] * fp16(up2 result[..., c]) - the level-4 output conv applied inside the resampler, modules.py:231
 * All pointers DEVICE fp32, NHWC maps, torch weight layouts (Cout,Cin,3,3) / side_w (Cout,Cin). */
typedef struct moge_test_conv_args {
    int32_t precision, B, H, W, Cin, Cout;
    int32_t relu_in, act, up2;               /* act: 0 none 1 relu */
    const float* x; const float* w; const float* bias;
    const float* add;                        /* (B,H,W,Cout) or NULL */
    const float* side; const float* side_w;  /* fused 1x1 side input (fp16, Cin == Cout) or NULL */
    const float* wu; const float* wv; float u0, u1, v0, v1;
    const float* w2; const float* bias2;     /* fused residual block (fp16, Cin == Cout == 64) or NULL */
    float* y;
    const float* dot_w; int32_t dot_rows;    /* up2 + fused 1x1 output conv (fp16, Cin 64, Cout 32): dot_w (dot_rows <= 4, 32) fp32; y is then (B,2H,2W,4) */
} moge_test_conv_args;
int moge_test_conv_ex(const moge_test_conv_args* args, void* stream);
/* ConvTranspose2d(k2, s2) + the 3x3 replicate-padded conv behind it (modules.py:160-165) through the fused path of the fp16 decoder (conv_pp.hip CT3: one composed
 * 4-phase conv on the low-res map + the border ring; Cin = 2 Cout, Cout 128 or 64; precision must be MOGE_FP16):
 *   y = conv3x3(convT(x; wt, bt); w3, b3) [+ side_w . side] [+ wu u + wv v at the output resolution]                 y, side (B,2H,2W,Cout)
 * no_border = 1 skips the border correction (tests: the interior must already be exact).  All pointers DEVICE fp32, NHWC maps, torch weight layouts. */
typedef struct moge_test_ct3_args {
    int32_t precision, B, H, W, Cin, Cout, no_border;
    const float* x; const float* wt; const float* bt; const float* w3; const float* b3;
    const float* side; const float* side_w;
    const float* wu; const float* wv; float u0, u1, v0, v1;
    float* y;
} moge_test_ct3_args;
int moge_test_ct3(const moge_test_ct3_args* args, void* stream);
/* ConvTranspose2d k2 s2, NHWC: x (B,H,W,Cin), w (Cin,Cout,2,2) torch layout, y (B,2H,2W,Cout) */
int moge_test_convt2x2(int precision, const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                       int Cin, int Cout, void* stream);
/* image (B,3,H,W) fp32 -> antialiased bilinear resize to (14*rows,14*cols), normalised, NCHW fp32 */
int moge_test_preprocess(const float* image, float* out, int B, int H, int W, int rows, int cols, void* stream);
/* MoGe-1 kernels: image (B,3,H,W) fp32 -> bicubic antialiased resize (OH,OW), NCHW fp32 (v1.py:275) */
int moge_test_resize_bicubic_aa(const float* image, float* out, int B, int H, int W, int OH, int OW, void* stream);
/* relu(GroupNorm(groups, C)(x)), eps 1e-5, NHWC x (B,H,W,C) fp32 in / out, computed in `precision` (v1.py:44-49) */
int moge_test_groupnorm_relu(int precision, const float* x, const float* gamma, const float* beta, float* y, int B, int H, int W, int C, int groups, void* stream);
/* act(norm(x)) of a v2 residual block (modules.py:31-58): groups = 0 no norm, 1 "layer_norm", C / 32 "group_norm", C "instance_norm" (gamma = beta = NULL);
 * act = moge_activation; in_place != 0 runs the kernel on its own input buffer (how the hidden norm of a block is applied) */
int moge_test_norm_act(int precision, const float* x, const float* gamma, const float* beta, float* y, int B, int H, int W, int C, int groups, int act, int in_place, void* stream);
/* pos_embed (1+37*37, D) -> (1+rows*cols, D) bicubic with the +0.1 kludge */
int moge_test_posembed(const float* pos, float* out, int D, int rows, int cols, void* stream);
/* focal/shift solve on (B,H,W,3) points + (B,H,W) 0/1 mask; focal_in NULL or (B,) */
int moge_test_recover(const float* points, const uint8_t* mask, const float* focal_in, int B, int H, int W,
                      float* focal, float* shift, int32_t* status, void* stream);

/* ---- optimal-alignment solvers of the evaluation path (reference: moge/utils/alignment.py, called by moge/test/metrics.py:128-282) -------
 * Stateless (no handle); every pointer is device memory; results are written asynchronously on `stream`.
 * moge_align_l1: alignment.py:52-89 with trunc=None - per row r: a[r] = argmin_a sum_i w[r,i] |a x[r,i] - y[r,i]|, loss[r] = that sum,
 * index[r] = the element whose ratio y/x the solution is.  1 <= n <= 15360 (a row is sorted inside one CU's LDS): otherwise MOGE_ERR_INVALID. */
int moge_align_l1(const float* x, const float* y, const float* w, int rows, int n, float eps, float* a, float* loss, int32_t* index, void* stream);
/* The anchor searches of alignment.py:163-212 (d = 1), :246-299 (d = 3, comp_mask 0b100) and :302-354 (d = 3, comp_mask 0b111) without their
 * (anchors, n, d) temporaries: src / tgt (B, n, d), weight (B, n); row r solves batch element row_batch[r] with sample row_anchor[r] subtracted
 * from the components selected by comp_mask (bit c = component c); outputs as moge_align_l1 over the n*d residuals (n*d <= 15360). */
int moge_align_l1_anchored(const float* src, const float* tgt, const float* weight, int n, int d, int comp_mask, const int32_t* row_batch,
                           const int32_t* row_anchor, int rows, float eps, float* scale, float* loss, int32_t* index, void* stream);
/* scatter_min of alignment.py:13-20 along dim 0: per batch element the minimum loss over its rows and the (last) row attaining it; -1 if none */
int moge_align_select(const float* loss, const int32_t* row_batch, int rows, int batch, float* min_loss, int32_t* min_row, void* stream);
/* alignment.py:399-415: per row the least-squares (a, b) of sqrt(w) x a + b ~ sqrt(w) y; w may be NULL (all ones); x, y, w are (rows, n) */
int moge_align_lstsq(const float* x, const float* y, const float* w, int rows, int n, float* a, float* b, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOGE_HIP_H */
