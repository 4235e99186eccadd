"""TEST INFRASTRUCTURE (runs in the build container only: imports /root/reference).  How wide is the REFERENCE's own fp16-vs-fp32 drift on one fixture when its rounding
noise is re-drawn?  A fixture's `drift16half` is ONE draw; outputs that are one number per image (the focal) or that inherit such a number (points / depth through the
recovered shift) move with it.  Here the unmodified reference runs `.half()` and fp32 on K copies of the fixture's image in which 1 % of the values are moved by ONE fp16 ulp
(the fp32 result moves by ~1e-5, the fp16 rounding noise of a 24-block network is a fresh draw) and prints the drift of every copy in the parity tests' metric.
    python oracle/reference_redraws.py v1_vitl_518 6 > profiles/r06an_reference_redraws_v1_vitl_518.log"""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden as G
from oracle import metrics as MX

def main():
    name, K = sys.argv[1], int(sys.argv[2])
    case = next(c for c in G.CASES if c["name"] == name)
    G.install_stubs()
    from moge.model import import_model_class_by_version
    OM = G.oracle_module(case)
    cfg = G.case_config(case)
    sd = G.case_state_dict(case, cfg)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "model.pt")
        OM.save_checkpoint(path, cfg, sd)
        model = import_model_class_by_version(case.get("version", "v2")).from_pretrained(path).eval()
    x0 = G.make_input(case)
    kw = dict(case["kwargs"]); kw16 = dict(kw); kw16["use_fp16"] = True
    G.install_half_stub()
    for d in range(K):
        x = x0.clone()
        if d > 0:
            g = torch.Generator().manual_seed(1000 + d)
            pick = torch.rand(x.shape, generator=g) < 0.01
            xh = x.half()
            up = torch.nextafter(xh.float(), torch.ones(())).half()          # (nextafter in fp32 then rounding may stay: step explicitly below)
            step = (xh.view(torch.int16) + 1).view(torch.float16)            # next fp16 value (positive inputs)
            x = torch.where(pick, step.float(), xh.float())
        model.float()
        ref = model.infer(x, **kw)
        model.half()
        out = {k: (v.float() if v.is_floating_point() else v) for k, v in model.infer(x, **kw16).items()}
        model.float()
        parts = []
        for k in ref:
            if ref[k].dtype == torch.bool:
                parts.append(f"mask flips {MX.mask_flips(out[k], ref[k]):.2e}")
            else:
                parts.append(f"{k} p999 {MX.summarize(k, out[k], ref[k])['p999']:.3e}")
        fx = lambda o: float(o["intrinsics"].reshape(-1, 3, 3)[0, 0, 0])
        print(f"{name} draw {d}: " + " | ".join(parts) + f" | focal fp32 {fx(ref):.6f} half {fx(out):.6f} rel {abs(fx(out) / fx(ref) - 1):.2e}", flush=True)

if __name__ == "__main__":
    main()
