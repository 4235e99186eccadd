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
