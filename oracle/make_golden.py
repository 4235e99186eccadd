"""
Golden-vector generator: runs the REAL reference (/root/reference, imported unmodified) on synthetic checkpoints
and seeded inputs and writes small fixtures to tests/golden/.  TEST INFRASTRUCTURE ONLY.

Run in the build container only (the GPU box has no /root/reference):
    python -m oracle.make_golden [--check-only]

The reference imports two packages that are not installed here (`utils3d`, `cv2`); both are stubbed before
import.  `cv2` is never called on the infer() path.  `utils3d.pt` gets the two functions infer() calls, with the
semantics restated in SURVEY.md 8(a16-a17) ("parity unpinned" boundary - utils3d is an un-vendored dependency
pinned at 3fab839f in the reference's pyproject.toml:23).

Every fixture records: the model config name, checkpoint seed, the input recipe, the infer() kwargs, the
reference outputs, a checksum of the synthetic weights and the torch/scipy versions that produced it.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

from . import moge_oracle as O
from . import moge_oracle_v1 as O1
from . import metrics as MX

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
REFERENCE_ROOT = "/root/reference"


def install_stubs():
    if "utils3d" in sys.modules:
        return
    sys.modules["cv2"] = types.ModuleType("cv2")
    u = types.ModuleType("utils3d")
    pt = types.ModuleType("utils3d.pt")
    npm = types.ModuleType("utils3d.np")

    def intrinsics_from_focal_center(fx, fy, cx, cy):
        fx, fy, cx, cy = torch.broadcast_tensors(fx, fy, cx, cy)
        K = torch.zeros(fx.shape + (3, 3), dtype=fx.dtype, device=fx.device)
        K[..., 0, 0] = fx
        K[..., 1, 1] = fy
        K[..., 0, 2] = cx
        K[..., 1, 2] = cy
        K[..., 2, 2] = 1
        return K

    def depth_map_to_point_map(depth, intrinsics=None, **_):
        H, W = depth.shape[-2:]
        u_ = (torch.arange(W, dtype=depth.dtype, device=depth.device) + 0.5) / W
        v_ = (torch.arange(H, dtype=depth.dtype, device=depth.device) + 0.5) / H
        fx, fy = intrinsics[..., 0, 0], intrinsics[..., 1, 1]
        cx, cy = intrinsics[..., 0, 2], intrinsics[..., 1, 2]
        x = (u_[None, :] - cx[..., None, None]) / fx[..., None, None] * depth
        y = (v_[:, None] - cy[..., None, None]) / fy[..., None, None] * depth
        return torch.stack([x, y, depth], dim=-1)

    pt.intrinsics_from_focal_center = intrinsics_from_focal_center
    pt.depth_map_to_point_map = depth_map_to_point_map
    u.pt, u.np = pt, npm
    sys.modules["utils3d"], sys.modules["utils3d.pt"], sys.modules["utils3d.np"] = u, pt, npm
    sys.path.insert(0, REFERENCE_ROOT)


def weights_digest(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].numpy().tobytes())
    return h.hexdigest()


def make_input(case: dict) -> torch.Tensor:
    """Seeded synthetic input (smooth-ish so the AA resize path is exercised on non-white noise)."""
    if case.get("input") == "house518":
        arr = np.load(os.path.join(GOLDEN_DIR, "house518_u8.npy"))
        return torch.from_numpy(arr.astype(np.float32) / 255.0).permute(2, 0, 1).contiguous()
    g = torch.Generator().manual_seed(case["input_seed"])
    shape = case["shape"]
    x = torch.rand(shape, generator=g)
    if case.get("input") == "rand":                 # exactly what bench.py / SURVEY 8(d) feed: torch.rand(..., generator=manual_seed(s))
        return x.contiguous()
    # add low-frequency structure: mix with a blurred copy
    xb = torch.nn.functional.avg_pool2d(x.reshape(-1, 3, *shape[-2:]), 5, 1, 2).reshape(shape)
    return (0.5 * x + 0.5 * xb).clamp(0, 1).contiguous()


CASES = [
    dict(name="tiny_b2_up", config="tiny-vits-normal", seed=0, sane=True, input_seed=1, shape=[2, 3, 98, 126],
         kwargs=dict(num_tokens=120, use_fp16=False)),
    dict(name="tiny_b1_down_3d", config="tiny-vits-normal", seed=0, sane=True, input_seed=2, shape=[3, 140, 150],
         kwargs=dict(num_tokens=56, use_fp16=False)),
    dict(name="tiny_fov_nomask_noproj", config="tiny-vits-normal", seed=0, sane=True, input_seed=3, shape=[2, 3, 84, 112],
         kwargs=dict(num_tokens=108, use_fp16=False, fov_x=55.0, apply_mask=False, force_projection=False)),
    dict(name="tiny_illposed", config="tiny-vits-normal", seed=1, sane=False, input_seed=4, shape=[2, 3, 96, 96],
         kwargs=dict(num_tokens=100, use_fp16=False)),
    dict(name="tiny_native37", config="tiny-vits-normal", seed=0, sane=True, input_seed=5, shape=[1, 3, 80, 80],
         kwargs=dict(num_tokens=1369, use_fp16=False)),
    dict(name="tiny_default_tokens_wide", config="tiny-vits-normal", seed=2, sane=True, input_seed=6, shape=[1, 3, 74, 148],
         kwargs=dict(use_fp16=False, resolution_level=0)),
    # onnx_compatible_mode = True (docs/onnx.md; v2.py:67-74): no antialiasing in the 14x resize, position embedding resampled by size -
    # on a down-sampling input (where AA matters) and on the native 37x37 grid (where the default mode bypasses the resampling)
    dict(name="tiny_onnx_mode_down", config="tiny-vits-normal", seed=0, sane=True, input_seed=7, shape=[2, 3, 140, 150], onnx=True,
         kwargs=dict(num_tokens=56, use_fp16=False)),
    dict(name="tiny_onnx_mode_native37", config="tiny-vits-normal", seed=0, sane=True, input_seed=8, shape=[1, 3, 80, 80], onnx=True,
         kwargs=dict(num_tokens=1369, use_fp16=False)),
    dict(name="vits_house518", config="moge-2-vits-normal", seed=0, sane=True, input="house518", shape=[3, 518, 518],
         kwargs=dict(use_fp16=False), stride=7),
    # BASELINE.json configs[1..4] at their own sizes: the image is torch.rand(seed) exactly as the bench draws it; default
    # resolution_level 9 -> num_tokens 3600 (v2.py:236-238); fixtures keep every 7th pixel.  B = 1: at B > 1 the CPU reference's own fp32 result depends on how ATen splits the batch
    # over threads (1e-5), which flips mask pixels that sit on the 0.5 threshold - the oracle could not be bit-compared with it.
    dict(name="vitb_normal_518_t3600", config="moge-2-vitb-normal", seed=0, sane=True, input="rand", input_seed=0, shape=[1, 3, 518, 518],
         kwargs=dict(use_fp16=False), stride=7),
    dict(name="vitl_518_t3600", config="moge-2-vitl", seed=0, sane=True, input="rand", input_seed=0, shape=[1, 3, 518, 518],
         kwargs=dict(use_fp16=False), stride=7),
    dict(name="vitl_normal_518x1036", config="moge-2-vitl-normal", seed=0, sane=True, input="rand", input_seed=0, shape=[1, 3, 518, 1036],
         kwargs=dict(use_fp16=False), stride=7),
    dict(name="vitl_normal_1036x518", config="moge-2-vitl-normal", seed=0, sane=True, input="rand", input_seed=1, shape=[1, 3, 1036, 518],
         kwargs=dict(use_fp16=False), stride=7),
    # the bench workload again with a DINOv2-like residual stream (oracle.add_massive_activations: three residual channels at -380 ... +600
    # and a common offset of 3 from block 2 on, all other channels O(1)): what the fp16 LayerNorm fold does when |x| / sigma is large
    dict(name="vitl_518_t3600_massive", config="moge-2-vitl", seed=0, sane=True, massive=True, input="rand", input_seed=0, shape=[1, 3, 518, 518],
         kwargs=dict(use_fp16=False), stride=7),
]
CASES += [
    # MoGe-1 (moge/model/v1.py): real v1 class on synthetic checkpoints
    dict(name="v1_tiny_b2", version="v1", config="tiny-v1-vits", seed=0, sane=True, input_seed=11, shape=[2, 3, 98, 126],
         kwargs=dict(num_tokens=120, use_fp16=False)),
    dict(name="v1_tiny_fov_nomask_noproj_3d", version="v1", config="tiny-v1-vits", seed=0, sane=True, input_seed=12, shape=[3, 84, 112],
         kwargs=dict(num_tokens=100, use_fp16=False, fov_x=60.0, apply_mask=False, force_projection=False)),
    dict(name="v1_tiny_default_tokens_wide", version="v1", config="tiny-v1-vits", seed=1, sane=True, input_seed=13, shape=[1, 3, 70, 140],
         kwargs=dict(use_fp16=False, resolution_level=3)),
    dict(name="v1_vitl_518", version="v1", config="moge-vitl", seed=0, sane=True, input="rand", input_seed=0, shape=[1, 3, 518, 518],
         kwargs=dict(use_fp16=False), stride=7),
]


def oracle_module(case: dict):
    return O1 if case.get("version") == "v1" else O


def case_state_dict(case: dict, cfg: dict):
    """The synthetic checkpoint of a case (every consumer - reference run, oracle, HIP tests - builds it through here)."""
    sd = oracle_module(case).synth_state_dict(cfg, case["seed"], case["sane"])
    if case.get("massive"):
        O.add_massive_activations(sd, cfg)
    return sd


# the cases whose reference run takes more than a few seconds on 8 cores (the CPU suite replays the oracle on the fast ones only)
SLOW_CASES = ("vits_house518", "vitb_normal_518_t3600", "vitl_518_t3600", "vitl_normal_518x1036", "vitl_normal_1036x518", "v1_vitl_518", "vitl_518_t3600_massive")


def run_reference(case: dict):
    install_stubs()
    from moge.model import import_model_class_by_version
    OM = oracle_module(case)
    MoGeModel = import_model_class_by_version(case.get("version", "v2"))
    cfg = OM.named_configs()[case["config"]]
    sd = case_state_dict(case, cfg)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "model.pt")
        OM.save_checkpoint(path, cfg, sd)
        model = MoGeModel.from_pretrained(path).eval()
    missing = set(model.state_dict().keys()) ^ set(sd.keys())
    assert not missing, f"state-dict key mismatch vs reference: {sorted(missing)[:8]}"
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
    x = make_input(case)
    if case.get("onnx"):
        model.onnx_compatible_mode = True
    out = model.infer(x, **case["kwargs"])
    fwd = model.forward(x if x.dim() == 4 else x[None], num_tokens=_tokens(cfg, case))
    # The reference's OWN fp16 path on the same input: fp32 weights + use_fp16=True = torch.autocast(float16) (v2.py:241; what
    # scripts/infer.py --fp16 / baselines/moge.py run when the model is not .half()).  It runs on CPU unmodified.  The other form,
    # model.half() (scripts/infer.py:84), does not: ATen has no Half kernel for the antialiased resize on CPU
    # ("compute_index_ranges_weights" not implemented for 'Half', modules.py:121), so it cannot be a fixture source here.
    kw16 = dict(case["kwargs"]); kw16["use_fp16"] = True
    out16 = model.infer(x, **kw16)
    return cfg, sd, x, out, fwd, out16


def _tokens(cfg, case):
    kw = case["kwargs"]
    if kw.get("num_tokens") is not None:
        return kw["num_tokens"]
    lo, hi = cfg["num_tokens_range"]
    return int(lo + (kw.get("resolution_level", 9) / 9) * (hi - lo))


def maxdiff(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double(), b.double()
    fin = torch.isfinite(a) & torch.isfinite(b)
    if not bool((torch.isfinite(a) == torch.isfinite(b)).all()):
        print(f"  !! non-finite pattern differs on {int((torch.isfinite(a) != torch.isfinite(b)).sum())} entries", flush=True)
    return float((a[fin] - b[fin]).abs().max()) if fin.any() else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true", help="compare oracle vs reference, write nothing")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    house = os.path.join(GOLDEN_DIR, "house518_u8.npy")
    if not os.path.exists(house):
        from PIL import Image
        im = Image.open(os.path.join(REFERENCE_ROOT, "example_images", "01_HouseIndoor.jpg")).convert("RGB")
        im = im.resize((518, 518), Image.BILINEAR)   # cv2 (INTER_AREA in the reference CLI) is not installed; PIL bilinear, recorded here
        np.save(house, np.asarray(im, dtype=np.uint8))
    import scipy
    for case in CASES:
        if args.only and case["name"] != args.only:
            continue
        cfg, sd, x, ref, ref_fwd, ref16 = run_reference(case)
        kw = {k: v for k, v in case["kwargs"].items() if k != "use_fp16"}
        tr = {}
        if case.get("version") == "v1":
            ora = O1.infer(cfg, sd, x, trace=tr, **kw)
        else:
            ora = O.infer(cfg, sd, x, trace=tr, onnx_compatible_mode=bool(case.get("onnx")), **kw)
        line = [case["name"]]
        assert set(ora.keys()) == set(ref.keys()), (ora.keys(), ref.keys())
        for k in ref:
            if ref[k].dtype == torch.bool:
                nd = int((ref[k] != ora[k]).sum())
                line.append(f"{k}:mismatch={nd}/{ref[k].numel()} true={float(ref[k].float().mean()):.3f}")
            else:
                line.append(f"{k}:{maxdiff(ref[k], ora[k]):.2e}")
        for k in ref_fwd:
            line.append(f"fwd.{k}:{maxdiff(ref_fwd[k], tr['forward'][k]):.2e}")
        line.append(f"focal={tr['focal'].tolist()} shift={tr['shift'].tolist()}")
        # drift of the reference's own fp16 (autocast) path against its fp32 path, in the per-pixel metric the parity tests use
        drift16 = {k: (dict(flips=MX.mask_flips(ref16[k], ref[k])) if ref[k].dtype == torch.bool else MX.summarize(k, ref16[k], ref[k])) for k in ref}
        line.append("ref-fp16 drift: " + " ".join(f"{k}:{(v.get('p999', v.get('flips'))):.2e}" for k, v in drift16.items()))
        print("  ".join(line), flush=True)
        if args.check_only:
            continue
        st = case.get("stride", 1)
        blob = {}
        for k, v in ref.items():
            a = v.numpy()
            if k != "intrinsics" and st > 1:
                a = a[..., ::st, ::st, :] if (a.ndim >= 3 and a.shape[-1] == 3 and k in ("points", "normal")) else a[..., ::st, ::st]
            blob["infer." + k] = a
        for k, v in ref16.items():          # the reference's fp16 (autocast) outputs, fp16 storage is enough for them
            a = v.numpy()
            if k != "intrinsics" and st > 1:
                a = a[..., ::st, ::st, :] if (a.ndim >= 3 and a.shape[-1] == 3 and k in ("points", "normal")) else a[..., ::st, ::st]
            blob["infer16." + k] = a
        for k, v in ref_fwd.items():
            a = v.detach().numpy()
            if st > 1 and k != "metric_scale":
                a = a[..., ::st, ::st, :] if (a.shape[-1] == 3 and k in ("points", "normal")) else a[..., ::st, ::st]
            blob["forward." + k] = a
        meta = dict(case=case, weights_sha256=weights_digest(sd), torch=torch.__version__, scipy=scipy.__version__,
                    numpy=np.__version__, threads=torch.get_num_threads(),
                    input_sha256=hashlib.sha256(x.numpy().tobytes()).hexdigest(),
                    focal=tr["focal"].tolist(), shift=tr["shift"].tolist(), drift16=drift16,
                    drift16_source="reference infer(use_fp16=True): fp32 weights under torch.autocast(cpu, float16), vs its own use_fp16=False output, full resolution")
        blob["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(GOLDEN_DIR, case["name"] + ".npz"), **blob)


if __name__ == "__main__":
    main()
