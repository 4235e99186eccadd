"""TEST INFRASTRUCTURE (runs in the build container only: imports /root/reference).  How wide is the REFERENCE's own fp16-vs-fp32 drift on one fixture when its rounding
noise is re-drawn?  A fixture's `drift16` / `drift16half` is ONE draw.  On most fixtures that does not matter (p99.9 over thousands of pixels is a stable statistic); on
a fixture whose focal / shift solve is ill-conditioned it does: the focal is one number per image, points and depth inherit the recovered shift, and ANY last-bit change
re-draws all three (v1_vitl_518: a 1-ulp change of 1 % of the input pixels moves the reference's fp32 focal by 1e-3 and its .half() drift between 1e-5 and 2.6e-3).
Here the unmodified reference runs fp32, autocast-fp16 and `.half()` on K copies of the fixture's image in which 1 % of the values are moved by ONE fp16 ulp (the fp16
rounding noise of a 24-block network is then a fresh draw; each copy's drift is against ITS OWN fp32 result) and writes every copy's drift in the parity tests' metric,
plus the drift of the raw point map in front of the solve (a stable statistic: mean |diff| / mean |points|), to tests/golden/<name>.redraws.json.  tests/golden_util.py
takes the band of such a fixture from the widest of these draws instead of from the first one.
    python oracle/reference_redraws.py v1_vitl_518 16 | tee profiles/r06an_reference_redraws_v1_vitl_518.log"""
import json, os, sys, tempfile
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
    nt = G._tokens(cfg, case)
    G.install_half_stub()
    fx = lambda o: float(o["intrinsics"].reshape(-1, 3, 3)[0, 0, 0])
    draws = []
    for d in range(K):
        xh = x0.half()
        x = xh.float()
        if d > 0:                       # draw 0 is the fixture's own image (rounded to fp16: what the .half() model sees)
            g = torch.Generator().manual_seed(1000 + d)
            pick = torch.rand(x.shape, generator=g) < 0.01
            step = (xh.view(torch.int16) + 1).view(torch.float16)            # the next fp16 value (inputs are >= 0)
            x = torch.where(pick, step.float(), x)
        xb = x if x.dim() == 4 else x[None]
        model.float()
        ref = model.infer(x, **kw)
        with torch.inference_mode():
            f32 = model.forward(xb, num_tokens=nt)["points"].float()
        rec = {"draw": d, "focal_fp32": fx(ref)}
        for form in ("autocast", "half"):
            if form == "half":
                model.half()
            out = {k: (v.float() if v.is_floating_point() else v) for k, v in model.infer(x, **kw16).items()}
            r = {}
            for k in ref:
                r[k] = float(MX.mask_flips(out[k], ref[k])) if ref[k].dtype == torch.bool else float(MX.summarize(k, out[k], ref[k])["p999"])
            r["focal"] = fx(out)
            if form == "half":
                with torch.inference_mode():
                    f16 = model.forward(xb.half(), num_tokens=nt)["points"].float()
                r["forward_points_noise"] = float((f16 - f32).abs().mean() / f32.abs().mean())
            model.float()
            rec[form] = r
            print(f"{name} draw {d} {form:8s}: " + " | ".join(f"{k} {v:.3e}" for k, v in r.items() if k != "focal") + f" | focal fp32 {fx(ref):.6f} fp16 {r['focal']:.6f} rel {abs(r['focal'] / fx(ref) - 1):.2e}", flush=True)
        draws.append(rec)
    dst = os.path.join(G.GOLDEN_DIR if hasattr(G, "GOLDEN_DIR") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"), name + ".redraws.json")
    with open(dst, "w") as f:
        json.dump({"case": name, "source": "oracle/reference_redraws.py: the unmodified reference on K copies of the fixture's image with 1 % of the values moved by one fp16 ulp; "
                                           "every copy's fp16 outputs against ITS OWN fp32 outputs, p99.9 of the per-pixel error (oracle/metrics.py), mask: fraction of flips; "
                                           "forward_points_noise: raw point map, mean |fp16 - fp32| / mean |fp32|", "draws": draws}, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
