"""How much does ONE fixture's fp16 error move when the attention noise is re-drawn?  The key-split attention (attn_pp16ks_kernel) sums the same fp32 terms in
another order and rounds its P operands against another running max: kernel-level error statistics are identical to the unsplit kernel's (tools/attn_ks_err.py),
but every split point is another draw.  Prints p99.9 / band per output for ATTN_KS = 0 and for several split points (ATTN_KS_MID), both fp16 forms.
    python tools/ks_split_draws.py v1_vitl_518 vitl_518_t3600"""
import sys, os, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import load_case, fp16_band, subsample
from oracle import metrics as MX
from moge_amd import _lib as L

def model_for(case):
    from moge_amd.model import import_model_class_by_version
    v1 = case.get("version") == "v1"
    if v1:
        from oracle import moge_oracle_v1 as O
    else:
        from oracle import moge_oracle as O
    cfg = O.named_configs()[case["config"]]
    sd = O.synth_state_dict(cfg, case["seed"], case["sane"])
    path = os.path.join(tempfile.mkdtemp(), "model.pt")
    O.save_checkpoint(path, cfg, sd)
    return import_model_class_by_version("v1" if v1 else "v2").from_pretrained(path).to("cuda").eval()

def main():
    for name in sys.argv[1:]:
        case, cfg, sd, x, gold, meta = load_case(name)
        model = model_for(case)
        st = case.get("stride", 1)
        g = {k[6:]: v for k, v in gold.items() if k.startswith("infer.")}
        kw = dict(case["kwargs"]); kw["use_fp16"] = True
        ntiles = None
        for form in ("autocast", "half"):
            band = fp16_band(meta, gold, form)
            for ks, mid in [(0, 0), (1, 0)] + [(1, m) for m in (3, 6, 9, 13, 16, 19, 25, 35, 45)]:
                L.tune("ATTN_KS", ks); L.tune("ATTN_KS_MID", mid)
                try:
                    out = (model.float() if form == "autocast" else model.half()).infer(x, **kw)
                finally:
                    model.float(); L.tune("ATTN_KS", 1); L.tune("ATTN_KS_MID", 0)
                parts = []
                for k in g:
                    a, b = subsample(k, out[k].cpu().numpy(), st), g[k]
                    if b.dtype == np.bool_:
                        parts.append(f"mask flips {(a != b).mean():.1e}/{band.get('mask', 0):.1e}")
                        continue
                    e, nmis, n = MX.pixel_errors(k, a, b)
                    if e.size:
                        parts.append(f"{k} {float(np.quantile(e, 0.999)) / band[k]:.2f}")
                print(f"{name} {form:8s} ATTN_KS={ks} MID={mid:2d}: " + " | ".join(parts), flush=True)

if __name__ == "__main__":
    main()
