"""GPU: a host WITHOUT Python-side compute and without torch - examples/host_without_torch.cpp, plain C against include/moge_hip.h - must produce
the bytes the Python mirror produces.  The drop-in boundary is the C ABI; this is the binding a C / C++ / Go / Rust host would write."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    exe = os.path.join(ROOT, "examples", "host_without_torch")
    src = exe + ".cpp"
    lib = os.path.join(ROOT, "moge_amd", "lib", "libmoge_hip.so")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        hipcc = "/opt/rocm/bin/hipcc"
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-w", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", os.path.dirname(lib), "-lmoge_hip",
                            "-Wl,-rpath,$ORIGIN/../moge_amd/lib"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    return exe


@pytest.mark.parametrize("config,prec", [("tiny-vits-normal", 0), ("tiny-vits-normal", 2), ("tiny-generic-stack", 1), ("tiny-block-options", 1)])
def test_c_host_without_torch_reproduces_the_python_mirror(tmp_path, config, prec):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moge_amd.model import import_model_class_by_version
    from oracle import moge_oracle as O
    exe = _build()
    cfg = O.named_configs()[config]
    ckpt = str(tmp_path / "model.pt")
    O.save_checkpoint(ckpt, cfg, O.synth_state_dict(cfg, 0, True))
    model = import_model_class_by_version("v2").from_pretrained(ckpt).to("cuda").eval()
    B, H, W, T = 2, 84, 112, 108
    rows, cols = model._grid(H, W, T)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(31))
    # what the C host reads: the config struct, the master blob, the image - plain bytes
    (tmp_path / "cfg.bin").write_bytes(bytes(model._cfg))
    (tmp_path / "master.blob").write_bytes(model.master_blob().cpu().numpy().tobytes())
    (tmp_path / "image.f32").write_bytes(x.numpy().tobytes())
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([exe, str(tmp_path / "cfg.bin"), str(tmp_path / "master.blob"), str(tmp_path / "image.f32"), str(B), str(H), str(W), str(rows), str(cols), str(prec),
                        str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    # the executable links libmoge_hip.so and the HIP runtime only
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libmoge_hip.so" in ldd and "torch" not in ldd and "python" not in ldd.lower()
    if prec == 2:
        model.half()
    ref = model.infer(x, num_tokens=T, use_fp16=prec != 0)
    model.float()
    raw = np.fromfile(tmp_path / "out.bin", dtype=np.uint8)
    px = B * H * W
    off = 0

    def take(nbytes, dtype, shape):
        nonlocal off
        a = raw[off:off + nbytes].view(dtype).reshape(shape)
        off += nbytes
        return a

    got = {"points": take(px * 12, np.float32, (B, H, W, 3)), "depth": take(px * 4, np.float32, (B, H, W)), "mask": take(px, np.uint8, (B, H, W)).astype(bool),
           "intrinsics": take(B * 36, np.float32, (B, 3, 3)), "normal": take(px * 12, np.float32, (B, H, W, 3))}
    assert off == raw.size
    for k, a in got.items():
        b = ref[k].cpu().numpy()
        if b.dtype == np.bool_:
            assert (a == b).all(), k
        else:
            assert np.array_equal(np.isfinite(a), np.isfinite(b)) and np.array_equal(a[np.isfinite(a)], b[np.isfinite(b)]), f"{k}: the C host's output differs from the Python mirror's"
