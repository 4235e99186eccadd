"""KS vs plain attention: error statistics against an fp64 reference (is the key-split form a different draw of the same noise, or worse?)."""
import torch, sys
sys.path.insert(0, ".")
from tests import hip_util as H
from moge_amd import _lib as L
import torch.nn.functional as F
torch.manual_seed(0)
for (B, nh, N, sink) in [(1, 16, 1370, 0.0), (1, 16, 1370, 8.0), (1, 16, 3601, 0.0), (1, 16, 3601, 8.0), (1, 6, 1370, 8.0)]:
    for seed in range(3):
        g = torch.Generator().manual_seed(seed)
        q, k, v = (torch.randn(B, nh, N, 64, generator=g) for _ in range(3))
        q = q * 1.5
        if sink:
            k[:, :, 0] = q.mean(dim=2) * 0 + torch.randn(B, nh, 64, generator=g) * sink   # a key with a large norm: some queries attend to it strongly
        ref = F.scaled_dot_product_attention(q.double().cuda(), k.double().cuda(), v.double().cuda()).permute(0, 2, 1, 3).reshape(B, N, nh * 64)
        # fp16-rounded inputs are what both kernels see
        qh, kh, vh = q.half().double().cuda(), k.half().double().cuda(), v.half().double().cuda()
        refh = F.scaled_dot_product_attention(qh, kh, vh).permute(0, 2, 1, 3).reshape(B, N, nh * 64)
        res = {}
        for ks in (0, 1, 2, 4):
            L.tune("ATTN_KS", ks)
            o = H.attention(1, q, k, v).double()
            res[ks] = o
        L.tune("ATTN_KS", 1)
        line = f"B{B} nh{nh} N{N} sink{sink} seed{seed}:"
        for ks, o in res.items():
            e = (o - refh)
            line += f"  ks{ks}: rms {e.pow(2).mean().sqrt().item():.3e} max {e.abs().max().item():.3e}"
        d = res[1] != res[0]
        line += f"  | ks1-ks0 rms {(res[1]-res[0]).pow(2).mean().sqrt().item():.3e}, {int(d.sum())} of {d.numel()} values differ"
        print(line, flush=True)
