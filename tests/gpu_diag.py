"""GPU diagnostic dump (not a test): per-kernel and per-stage errors in one go, printed and written to gpurun_out/."""
import json, os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tests import hip_util as H
from tests.golden_util import load_case, rel_err
from oracle import moge_oracle as O

out = {}
def rec(name, fn):
    try:
        t = time.time(); v = fn(); out[name] = v; print(f"{name:50s} {v}  ({time.time()-t:.1f}s)", flush=True)
    except Exception as e:
        out[name] = "EXC " + repr(e); print(f"{name:50s} EXC {e!r}", flush=True); traceback.print_exc()

def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))

g = torch.Generator().manual_seed(0)
for prec in (0, 1):
    for (M, N, K) in [(128, 128, 64), (300, 256, 192), (129, 64, 72), (77, 32, 40)]:
        A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
        rec(f"gemm p{prec} {M}x{N}x{K}", lambda: relmax(H.gemm(prec, A, W, b, 0), A.cuda() @ W.cuda().T + b.cuda()))
    x = torch.randn(100, 384, generator=g); w = torch.randn(384, generator=g); bb = torch.randn(384, generator=g)
    rec(f"layernorm p{prec}", lambda: relmax(H.layernorm(prec, x, w, bb), F.layer_norm(x, (384,), w, bb, 1e-6)))
    for N in (64, 130, 1370):
        q, k, v = (torch.randn(1, 2, N, 64, generator=g) for _ in range(3))
        rec(f"attention p{prec} N={N}", lambda: relmax(H.attention(prec, q, k, v), F.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(1, N, 128)))
    xc = torch.randn(2, 32, 13, 17, generator=g); wc = torch.randn(64, 32, 3, 3, generator=g) / 17; bc = torch.randn(64, generator=g)
    rec(f"conv3x3 p{prec}", lambda: relmax(H.conv3x3(prec, xc.permute(0, 2, 3, 1), wc, bc), F.conv2d(F.pad(xc, (1, 1, 1, 1), mode="replicate"), wc, bc).permute(0, 2, 3, 1)))
    rec(f"conv3x3 up2 p{prec}", lambda: relmax(H.conv3x3(prec, xc.permute(0, 2, 3, 1), wc, bc, up2=True), F.conv2d(F.pad(F.interpolate(xc, scale_factor=2, mode="bilinear"), (1, 1, 1, 1), mode="replicate"), wc, bc).permute(0, 2, 3, 1)))
    wt = torch.randn(32, 64, 2, 2, generator=g) / 6
    rec(f"convT p{prec}", lambda: relmax(H.convt2x2(prec, xc.permute(0, 2, 3, 1), wt, bc), F.conv_transpose2d(xc, wt, bc, stride=2).permute(0, 2, 3, 1)))

img = torch.rand(1, 3, 98, 126, generator=g)
mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
rec("preprocess up", lambda: float((H.preprocess(img, 10, 12).cpu() - (F.interpolate(img, (140, 168), mode="bilinear", antialias=True) - mean) / std).abs().max()))
rec("preprocess down", lambda: float((H.preprocess(img, 5, 6).cpu() - (F.interpolate(img, (70, 84), mode="bilinear", antialias=True) - mean) / std).abs().max()))
pos = torch.randn(1, 1370, 384, generator=g)
rec("posembed", lambda: float((H.posembed(pos[0], 10, 12).cpu() - O.pos_embed_for_grid(pos, 10, 12)[0]).abs().max()))

# end to end, stage by stage
from moge_amd.model import import_model_class_by_version
MoGeModel = import_model_class_by_version("v2")
case, cfg, sd, x, gold, meta = load_case("tiny_b2_up")
O.save_checkpoint("/tmp/tiny.pt", cfg, sd)
model = MoGeModel.from_pretrained("/tmp/tiny.pt").to("cuda").eval()
tr = {}
ref = O.forward(cfg, sd, x, 120, tr)
for prec_name, m in (("fp32", model.float()), ("fp16", model.half())):
    try:
        fwd = m.forward(x, 120)
        taps = torch.cat([t[:, 1:] for t in tr["taps"]], dim=-1).reshape(-1)
        rec(f"{prec_name} tapcat", lambda: rel_err(m.debug_tap("tapcat").cpu().numpy(), taps.numpy()))
        rec(f"{prec_name} cls", lambda: rel_err(m.debug_tap("cls").cpu().numpy(), tr["cls"].reshape(-1).numpy()))
        rec(f"{prec_name} features", lambda: rel_err(m.debug_tap("features").cpu().numpy(), tr["features"].permute(0, 2, 3, 1).reshape(-1).numpy()))
        for l, n in enumerate(tr["neck"]):
            rec(f"{prec_name} neck{l}", lambda: rel_err(m.debug_tap(f"neck{l}").cpu().numpy(), n.permute(0, 2, 3, 1).reshape(-1).numpy()))
        for k in ref:
            rec(f"{prec_name} forward.{k}", lambda: rel_err(fwd[k].float().cpu().numpy(), ref[k].numpy()))
    except Exception as e:
        print("forward failed", prec_name, repr(e)); traceback.print_exc()
model.float()
oref = O.infer(cfg, sd, x, num_tokens=120)
for use_fp16 in (False, True):
    try:
        o = model.infer(x, num_tokens=120, use_fp16=use_fp16)
        for k in oref:
            if oref[k].dtype == torch.bool:
                rec(f"infer fp16={use_fp16} {k} mismatches", lambda: int((o[k].cpu() != oref[k]).sum()))
            else:
                a, b = o[k].cpu().numpy(), oref[k].numpy()
                import numpy as np
                fin = np.isfinite(a) & np.isfinite(b)
                rec(f"infer fp16={use_fp16} {k}", lambda: float((np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1)).max()))
    except Exception as e:
        print("infer failed", repr(e)); traceback.print_exc()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/diag.json", "w"), indent=1)
