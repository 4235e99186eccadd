"""Steady-state PCIe-inclusive rate of moge_amd.pipeline.InferPipeline (vitl, B=32, 518x518, fp16) vs resident-input infer()."""
import os, sys, time, tempfile
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moge_amd.model import import_model_class_by_version
from moge_amd.pipeline import InferPipeline
from oracle import moge_oracle as O

cfg = O.named_configs()["moge-2-vitl"]
sd = O.synth_state_dict(cfg, 0, True)
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "m.pt"); O.save_checkpoint(p, cfg, sd)
    model = import_model_class_by_version("v2").from_pretrained(p).to("cuda").eval().half()
B = 32
u8 = np.random.default_rng(0).integers(0, 256, size=(B, 518, 518, 3), dtype=np.uint8)
xd = torch.from_numpy(u8).cuda()
for _ in range(2): model.infer_uint8(xd)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(6): model.infer_uint8(xd)
torch.cuda.synchronize(); print("resident uint8 infer: %.1f img/s" % (6 * B / (time.perf_counter() - t)))
for slots in (2, 3):
    for keys in (None, ("depth", "mask", "intrinsics")):
        pipe = InferPipeline(model, B, 518, 518, outputs=keys, slots=slots, use_fp16=True)
        for _ in pipe.run(iter([u8] * 2), copy=False): pass
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 10
        for _ in pipe.run(iter([u8] * n), copy=False): pass
        dt = time.perf_counter() - t
        print("pipeline slots=%d outputs=%s: %.1f img/s" % (slots, keys or "all", n * B / dt))
        del pipe

# steady-state slope (excludes the pipeline's fill / drain): time of n batches for n = 6, 12, 24 in one pipeline object
pipe = InferPipeline(model, B, 518, 518, slots=2, use_fp16=True)
for _ in pipe.run(iter([u8] * 3), copy=False): pass
ts = {}
for n in (6, 12, 24):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in pipe.run(iter([u8] * n), copy=False): pass
    ts[n] = time.perf_counter() - t
    print("pipeline slots=2 all outputs, %2d batches: %.1f img/s  (%.1f ms per batch)" % (n, n * B / ts[n], ts[n] / n * 1e3))
print("steady-state slope 12 -> 24 batches: %.1f img/s (%.1f ms per batch); fill + drain = %.1f ms" % (
    12 * B / (ts[24] - ts[12]), (ts[24] - ts[12]) / 12 * 1e3, (ts[12] - 12 * (ts[24] - ts[12]) / 12) * 1e3))
# where a batch's time goes on the host side of one submission
t = time.perf_counter(); pipe.pin_in[0].copy_(torch.from_numpy(u8)); print("host memcpy into the pinned slot: %.2f ms" % ((time.perf_counter() - t) * 1e3))
torch.cuda.synchronize(); t = time.perf_counter(); pipe.dev_in[0].copy_(pipe.pin_in[0], non_blocking=True); torch.cuda.synchronize(); print("H2D 25.8 MB: %.2f ms" % ((time.perf_counter() - t) * 1e3))
out = model.infer_uint8(pipe.dev_in[0]); torch.cuda.synchronize()
t = time.perf_counter()
for k in pipe.keys: pipe.pin_out[0][k].copy_(out[k], non_blocking=True)
torch.cuda.synchronize(); print("D2H of all outputs (%.0f MB): %.2f ms" % (sum(out[k].numel() * out[k].element_size() for k in pipe.keys) / 1e6, (time.perf_counter() - t) * 1e3))
