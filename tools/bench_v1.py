"""Throughput of the MoGe-1 path (moge_amd.model.v1.MoGeModel.infer) on one GPU, synthetic checkpoint: not the headline metric (bench.py is MoGe-2,
BASELINE.json), a record of where the untuned v1 decoder stands.   python tools/bench_v1.py [--config moge-vitl-train-config] [--batch 32] [--steps 5]"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="moge-vitl-train-config")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=518)
    args = ap.parse_args()
    from moge_amd.model import import_model_class_by_version
    from oracle import moge_oracle_v1 as O1            # synthetic checkpoint generator only
    cfg = O1.named_configs()[args.config]
    path = os.path.join(tempfile.mkdtemp(), "model.pt")
    O1.save_checkpoint(path, cfg, O1.synth_state_dict(cfg, 0, True))
    model = import_model_class_by_version("v1").from_pretrained(path).to("cuda").eval().half()
    x = torch.rand(args.batch, 3, args.size, args.size, generator=torch.Generator().manual_seed(0)).cuda()
    x1 = x[:1].contiguous()
    res = {"workload": f"{args.config} infer(): batch {args.batch} x 3x{args.size}x{args.size}, default num_tokens ({cfg['num_tokens_range'][1]}), model.half(), synthetic checkpoint"}
    for name, inp, n in (("images_per_s", x, args.batch), ("batch1_ms", x1, 1)):
        for _ in range(args.warmup):
            model.infer(inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps if n > 1 else 20):
            model.infer(inp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = round(n * args.steps / dt, 2) if n > 1 else round(dt / 20 * 1e3, 3)
    model.profile(True)
    model.profile_read(reset=True)
    model.infer(x)
    torch.cuda.synchronize()
    prof = model.profile_read(reset=True)
    model.profile(False)
    res["kernel_classes_ms"] = {k: round(v["ms"], 2) for k, v in prof.items() if v.get("ms", 0) > 0.005}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
