// A host of libmoge_hip.so that uses NO Python and NO torch: plain C ABI (include/moge_hip.h) + the HIP runtime for the buffers it owns.
// It is the skeleton a C / C++ / Go (cgo) / Rust (FFI) service would follow, and - with the RCCL block below - a rank of the multi-GPU deployment
// of SURVEY.md 8(e) without torch.distributed.
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/host_without_torch.cpp -o examples/host_without_torch -L moge_amd/lib -lmoge_hip \
//         -Wl,-rpath,'$ORIGIN/../moge_amd/lib'
//   examples/host_without_torch cfg.bin master.blob image.f32 B H W token_rows token_cols precision out.bin
//
//   cfg.bin      the bytes of a `moge_config` (what MoGeModel.__init__ derives from the checkpoint's model_config; moge_amd/model/v2.py fills the
//                same struct through ctypes)
//   master.blob  the fp32 master weight blob in the library's layout (MoGeModel.save_blob payload: order = f(config) only) - one H2D copy
//   image.f32    B x 3 x H x W float32 in [0, 1]
//   out.bin      points (B,H,W,3) f32 | depth (B,H,W) f32 | mask (B,H,W) u8 | intrinsics (B,3,3) f32 | normal (B,H,W,3) f32   (absent heads are skipped)
//
// tests/test_hip_host_example.py builds it, runs it on files written by the Python mirror and compares out.bin with MoGeModel.infer() bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "moge_hip.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define MOGEOK(x) do { int e_ = (x); if (e_ != 0) { fprintf(stderr, "%s: status %d: %s\n", #x, e_, moge_last_error()); return 3; } } while (0)

static std::vector<char> read_file(const char* path) {
    std::vector<char> v;
    FILE* f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (n > 0 && fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc != 11) {
        fprintf(stderr, "usage: %s cfg.bin master.blob image.f32 B H W token_rows token_cols precision(0 fp32, 1 autocast fp16, 2 half) out.bin\n", argv[0]);
        return 1;
    }
    const int B = atoi(argv[4]), H = atoi(argv[5]), W = atoi(argv[6]), rows = atoi(argv[7]), cols = atoi(argv[8]), prec = atoi(argv[9]);
    const std::vector<char> cfgb = read_file(argv[1]), blob = read_file(argv[2]), img = read_file(argv[3]);
    if (cfgb.size() != sizeof(moge_config)) { fprintf(stderr, "cfg.bin has %zu bytes, moge_config has %zu (ABI %d)\n", cfgb.size(), sizeof(moge_config), MOGE_ABI_VERSION); return 1; }
    if (img.size() != (size_t)B * 3 * H * W * 4) { fprintf(stderr, "image.f32 size mismatch\n"); return 1; }
    if (moge_abi_version() != MOGE_ABI_VERSION) { fprintf(stderr, "library ABI %d, header %d\n", moge_abi_version(), MOGE_ABI_VERSION); return 1; }
    moge_config cfg;
    memcpy(&cfg, cfgb.data(), sizeof(cfg));

    HIPOK(hipSetDevice(0));
    hipStream_t st;
    HIPOK(hipStreamCreate(&st));
    moge_handle* h = nullptr;
    MOGEOK(moge_create(&cfg, 0, &h));                            // MoGeModel.__init__                 (moge/model/v2.py:30-57)
    // weights: the master blob as ONE copy (moge_load_weights with a tensor list is the other way in, for a host that reads the checkpoint itself)
    MOGEOK(moge_alloc_master(h));
    void* master = nullptr;
    size_t master_bytes = 0;
    MOGEOK(moge_master_blob(h, &master, &master_bytes));
    if (blob.size() != master_bytes) { fprintf(stderr, "master.blob has %zu bytes, this config needs %zu\n", blob.size(), master_bytes); return 1; }
    HIPOK(hipMemcpy(master, blob.data(), master_bytes, hipMemcpyHostToDevice));
    MOGEOK(moge_master_ready(h));
    // (a rank of an N-GPU job would instead do, with its own ncclComm_t `comm` and only rank `root` holding the blob:
    //      MOGEOK(moge_broadcast_weights(h, comm, root, st));        one ncclBroadcast over xGMI, then every rank infers its own shard)
    MOGEOK(moge_set_precision(h, prec, st));                     // .float() / autocast / .half()            (scripts/infer.py:82-84, v2.py:241)

    const size_t px = (size_t)B * H * W;
    void* d_img = nullptr;
    float *d_pts = nullptr, *d_dep = nullptr, *d_nrm = nullptr, *d_K = nullptr;
    uint8_t* d_msk = nullptr;
    HIPOK(hipMalloc(&d_img, img.size()));
    HIPOK(hipMemcpyAsync(d_img, img.data(), img.size(), hipMemcpyHostToDevice, st));
    moge_outputs out;
    memset(&out, 0, sizeof(out));
    if (cfg.heads & MOGE_HEAD_POINTS) {
        HIPOK(hipMalloc(&d_pts, px * 12)); HIPOK(hipMalloc(&d_dep, px * 4)); HIPOK(hipMalloc(&d_K, (size_t)B * 36));
        out.points = d_pts; out.depth = d_dep; out.intrinsics = d_K;
    }
    if (cfg.heads & MOGE_HEAD_MASK) { HIPOK(hipMalloc(&d_msk, px)); out.mask = d_msk; }
    if (cfg.heads & MOGE_HEAD_NORMAL) { HIPOK(hipMalloc(&d_nrm, px * 12)); out.normal = d_nrm; }
    // img_dtype 0 = fp32 image as it is; 3 = fp32 values rounded to fp16 as they are read: `image.to(dtype=self.dtype)` of a .half() model (v2.py:229)
    MOGEOK(moge_infer(h, d_img, prec == MOGE_FP16_HALF ? 3 : 0, B, H, W, rows, cols, /*fov_x*/ nullptr, MOGE_FORCE_PROJECTION | MOGE_APPLY_MASK, &out, st));   // MoGeModel.infer (v2.py:194-303)
    MOGEOK(moge_sync(h, st));                                    // MOGE_ERR_NONFINITE = scipy's ValueError in the reference

    FILE* f = fopen(argv[10], "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", argv[10]); return 1; }
    std::vector<char> host(px * 12);
    auto dump = [&](const void* dptr, size_t bytes) -> int {
        if (!dptr) return 0;
        if (host.size() < bytes) host.resize(bytes);
        if (hipMemcpy(host.data(), dptr, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        return fwrite(host.data(), 1, bytes, f) == bytes ? 0 : 1;
    };
    int bad = dump(d_pts, px * 12) | dump(d_dep, px * 4) | dump(d_msk, px) | dump(d_K, (size_t)B * 36) | dump(d_nrm, px * 12);
    fclose(f);
    moge_destroy(h);
    (void)hipFree(d_img); (void)hipFree(d_pts); (void)hipFree(d_dep); (void)hipFree(d_nrm); (void)hipFree(d_K); (void)hipFree(d_msk);
    (void)hipStreamDestroy(st);
    if (bad) { fprintf(stderr, "writing the outputs failed\n"); return 1; }
    printf("host_without_torch: infer ok (B=%d %dx%d, grid %dx%d, precision %d)\n", B, H, W, rows, cols, prec);
    return 0;
}
