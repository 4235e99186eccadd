"""Vendor reference points on the same box, same shapes, same random data (tools only - nothing here is on the product path):
torch's fused SDPA (ROCm flash attention) and MIOpen convolutions for the hot shapes of moge-2-vitl B=32, beside the library's own kernels.
    python tools/vendor_refpoints.py > gpurun_out/vendor_refpoints.json"""
import json
import time

import torch
import torch.nn.functional as F


def bench(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = "cuda"
    out = {"torch": torch.__version__, "device": torch.cuda.get_device_name(0)}
    g = torch.Generator(device=dev).manual_seed(0)
    B, nh, N, D = 32, 16, 3601, 64
    q, k, v = (torch.randn(B, nh, N, D, device=dev, dtype=torch.float16, generator=g) for _ in range(3))
    fl = 4.0 * B * nh * N * N * D
    res = {}
    for name, backend in (("flash", "FLASH_ATTENTION"), ("efficient", "EFFICIENT_ATTENTION"), ("default", None)):
        try:
            if backend is None:
                t = bench(lambda: F.scaled_dot_product_attention(q, k, v))
            else:
                from torch.nn.attention import SDPBackend, sdpa_kernel
                with sdpa_kernel(getattr(SDPBackend, backend)):
                    t = bench(lambda: F.scaled_dot_product_attention(q, k, v))
            res[name] = {"ms": round(t * 1e3, 3), "tflops": round(fl / t / 1e12, 1)}
        except Exception as e:          # noqa: BLE001
            res[name] = {"error": str(e)[:200]}
    out["sdpa_fp16_B32_h16_N3601_D64"] = res
    convs = {}
    for name, (Bc, C, H, W) in {"64->64@480": (32, 64, 480, 480), "128->128@240": (32, 128, 240, 240), "256->256@120": (32, 256, 120, 120)}.items():
        x = torch.randn(Bc, C, H, W, device=dev, dtype=torch.float16, generator=g).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(C, C, 3, 3, device=dev, dtype=torch.float16, generator=g) / (9 * C) ** 0.5).contiguous(memory_format=torch.channels_last)
        b = torch.randn(C, device=dev, dtype=torch.float16, generator=g)
        try:
            t = bench(lambda: F.conv2d(x, w, b, padding=1))
            convs[name] = {"ms": round(t * 1e3, 3), "tflops": round(18.0 * Bc * H * W * C * C / t / 1e12, 1), "note": "zero padding (MIOpen has no replicate mode), NHWC fp16"}
        except Exception as e:          # noqa: BLE001
            convs[name] = {"error": str(e)[:200]}
    out["conv3x3_fp16_nhwc"] = convs
    gm = {}
    for name, (M, N_, K) in {"qkv": (115232, 3072, 1024), "proj": (115232, 1024, 1024), "fc1": (115232, 4096, 1024), "fc2": (115232, 1024, 4096)}.items():
        a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
        w = torch.randn(N_, K, device=dev, dtype=torch.float16, generator=g)
        t = bench(lambda: a @ w.T)
        gm[name] = {"ms": round(t * 1e3, 3), "tflops": round(2.0 * M * N_ * K / t / 1e12, 1), "note": "torch.matmul fp16 (hipBLASLt / rocBLAS), plain store, no epilogue"}
    out["gemm_fp16"] = gm
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
