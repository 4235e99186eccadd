// Host-callable launchers of every kernel in libmoge_hip.so (definitions in the .hip files).
#pragma once
#include "common.h"

template <typename T> int launch_attention(const void* q, const void* k, const void* vT, void* out, int B, int nh, int Ntok, int Npad, hipStream_t st);

// fp16 throughput path (attention_pp.hip): q, k, v all (B, nh, Ntok, 64)
int launch_attention_pp(const void* q, const void* k, const void* v, void* out, int B, int nh, int Ntok, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0);
size_t attention_pp_ws_bytes(int B, int nh, int Ntok);            // stream-K form (sub-round grids): workspace wanted, 0 = not taken for this shape
size_t attention_pp_ws_counter_bytes(int B, int nh, int Ntok);    // ... of which this many leading bytes must be zero before the first launch (the kernel leaves them zero)

template <typename TIn, typename TOut>
int launch_preprocess(const void* img, void* out, int B, int H, int W, int rows, int cols, int ldk, int nchw_out, int round16, int aa, const float* mean,
                      const float* std_, hipStream_t st, int* zero_i32 = nullptr, int zero_n = 0);      // patch layout: also zeroes the K padding columns [588, ldk) and zero_i32[0 .. zero_n)
// LN fold (fp16 path): fp16 copy + (mean, rstd) of the first block's input; partial sums -> (mean, rstd); weight folding at pack time
template <typename T> int launch_ln_raw(const float* x, void* out, float* mr, long rowsN, int D, hipStream_t st);
int launch_ln_finalize(const float* part, float* mr, long rowsN, int NP, int D, hipStream_t st);
template <typename T> int launch_fold_ln(const float* W, const float* g, const float* beta, const float* b, void* Wf, float* c, float* bf, int N, int K, hipStream_t st);
template <typename T> int launch_u8hwc_to_chw(const void* in, void* out, int B, int H, int W, hipStream_t st);      // uint8 (B,H,W,3) -> T (B,3,H,W), /255
int launch_depth_edge_mask(const float* depth, const unsigned char* mask, unsigned char* out, int B, int H, int W, float rtol, hipStream_t st);
int launch_posembed(const float* pos, float* out, int D, int rows, int cols, int size_mode, hipStream_t st);
template <typename T>
int launch_layernorm(const float* x, const float* w, const float* b, void* out, float* cls_out, long rowsN, int D, int ldo, int coloff,
                     int tap_mode, int Ntok, hipStream_t st);
int launch_layernorm_x16(const void* x16, const float* w, const float* b, void* out, float* cls_out, long rowsN, int D, int ldo, int coloff,
                         int tap_mode, int Ntok, hipStream_t st);        // fp16 residual stream in, fp16 out (cls_out fp32)
template <typename TS, typename TD> int launch_convert(const void* s, void* d, long n, hipStream_t st);
template <typename TD>
int launch_repack(const float* src, void* dst, int n0, int n1, int n2, int n3, long ss0, long ss1, long ss2, long ss3, long ds0, long ds1,
                  long ds2, hipStream_t st);

// CT3 (conv_pp.hip): composed weights of ConvTranspose2d + 3x3 from the fp32 torch-layout weights (model.hip): wc f16 [4 co][4 ci], dw f16 [12][co][2 ci]
int ct3_compose_device(const float* w3, const float* wt, int ci, int co, void* wc, void* dw, hipStream_t st);
template <typename T> int launch_pack_phase_conv(const float* w, void* dst, int Cout, int Cin, hipStream_t st, int nearest = 0);      // nearest: x2 nearest instead of bilinear
int launch_pack_dot_table(const float* w, int rows, int nd, void* tab, hipStream_t st);      // GemmArgs::dot_tab from fp32 rows [rows][32]
// head_final on the fused output-conv maps (conv_pp.hip DOT): y (B,Hd,Wd,4) fp32 = Wo . x4, z (B,Hd,Wd,zld) fp32 with this head's group at
// channel zoff (= (Wo . Win4) . neck4), or null; out = remap(resize(y + z) + bias)
int launch_head_final_dot(int kind, const float* y, const float* z, int zld, int zoff, const float* bias, float* out, int B, int Hd, int Wd, int H, int W,
                          int remap, hipStream_t st);

template <typename T>
int launch_head_final_k3(int kind, const void* x4, const float* w, const float* bias, float* out, int B, int Hd, int Wd, int C, int H, int W, int remap, hipStream_t st);
template <typename T>
int launch_head_final(int kind, const void* x4, const float* w, const float* bias, const void* n4, const float* w2, float* out, int B, int Hd,
                      int Wd, int C, int H, int W, int remap, hipStream_t st, int ld = 0);      // ld: channel pitch of x4 / n4 (0 = C)
int launch_mlp_layer(const float* in, const float* W, const float* bias, float* out, int B, int K, int N, int act, hipStream_t st);
int launch_recover(const float* points, const float* mask_prob, const uint8_t* mask_u8, const float* fov_deg, const float* focal_in, int B,
                   int H, int W, float* focal, float* shift, float* intrinsics, int* status, hipStream_t st, float mask_thr = 0.5f);
int launch_finalize(const float* points_in, const float* normal_in, const float* mask_prob, const float* metric, const float* shift,
                    const float* intr, int B, int H, int W, int flags, float* points_out, float* depth_out, float* normal_out,
                    uint8_t* mask_out, hipStream_t st, float mask_thr = 0.5f);

// ---- MoGe-1 (moge/model/v1.py) support kernels (elementwise.hip) ----
template <typename TIn> int launch_resize_bicubic_aa(const void* img, float* out, int B, int H, int W, int OH, int OW, int round16, hipStream_t st);
template <typename T>
int launch_groupnorm_relu(const void* x, void* y, const float* gamma, const float* beta, float* scratch, int B, int H, int W, int C, int G, hipStream_t st);
template <typename T>
int launch_groupnorm_act(const void* x, void* y, const float* gamma, const float* beta, float* scratch, int B, int H, int W, int C, int G, int act, hipStream_t st);
size_t groupnorm_scratch_floats(int B, int H, int W, int G);
template <typename T>
int launch_resize_bilinear_uv(const void* x, void* out, int B, int hs, int ws, int C, int OH, int OW, int Cp, float u0, float u1, float v0, float v1, hipStream_t st);
