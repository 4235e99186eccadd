"""Where does a last-bit change of the attention outputs get amplified - in the network or in the focal / shift solve behind it?  For one fixture (.half() form): raw forward()
outputs and infer() outputs with the batch-invariant attention (ATTN_KS = 0) against the key-split form at several split points; every row is relative to ATTN_KS = 0.
    python tools/ks_forward_sensitivity.py v1_vitl_518 2500"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import load_case
from tools.ks_split_draws import model_for
from moge_amd import _lib as L

name, nt = sys.argv[1], int(sys.argv[2])
case, cfg, sd, x, gold, meta = load_case(name)
model = model_for(case).half()
kw = dict(case["kwargs"]); kw["use_fp16"] = True
xb = x if x.dim() == 4 else x[None]
def run(ks, mid):
    L.tune("ATTN_KS", ks); L.tune("ATTN_KS_MID", mid)
    try:
        f = {k: v.float().cpu().numpy() for k, v in model.forward(xb, nt).items()}
        o = {k: v.float().cpu().numpy() for k, v in model.infer(x, **kw).items()}
    finally:
        L.tune("ATTN_KS", 1); L.tune("ATTN_KS_MID", 0)
    return f, o
f0, o0 = run(0, 0)
g = {k[6:]: v for k, v in gold.items() if k.startswith("infer.")}
st = case.get("stride", 1)
gf = gold["forward.points"].astype(np.float64)
def fwd_noise(f):          # the library's raw point map against the reference's fp32 forward (strided fixture): mean |diff| / mean |points|
    return float(np.abs(f["points"][:, ::st, ::st].astype(np.float64) - gf).mean() / np.abs(gf).mean())
print(f"{name} ATTN_KS=0: forward points vs the fp32 reference: mean |diff| / mean |points| {fwd_noise(f0):.2e}", flush=True)
for ks, mid in [(1, 0), (1, 9), (1, 16), (1, 35)]:
    f, o = run(ks, mid)
    fp = np.abs(f["points"] - f0["points"]); sc = np.abs(f0["points"]).mean()
    fin = np.isfinite(o["depth"]) & np.isfinite(o0["depth"])
    dd = np.abs(o["depth"][fin] / o0["depth"][fin] - 1)
    print(f"{name} ATTN_KS={ks} MID={mid:2d}: forward vs fp32 reference {fwd_noise(f):.2e} | forward points |diff| mean {fp.mean() / sc:.2e} p99.9 {np.quantile(fp, 0.999) / sc:.2e} max {fp.max() / sc:.2e} (of mean |points|)"
          f" | infer depth rel diff mean {dd.mean():.2e} p99.9 {np.quantile(dd, 0.999):.2e} | focal {o['intrinsics'].reshape(-1, 3, 3)[0, 0, 0]:.6f} vs {o0['intrinsics'].reshape(-1, 3, 3)[0, 0, 0]:.6f} (fp32 reference {g['intrinsics'].reshape(-1, 3, 3)[0, 0, 0]:.6f})", flush=True)
