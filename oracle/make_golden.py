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

Three reference runs per case: fp32 (`infer.*`), fp32 weights + use_fp16=True = torch.autocast(float16) (`infer16.*`) and `model.half()`
(`infer16half.*`, scripts/infer.py:83-84).  The last one needs ONE more documented stub on CPU: ATen has no Half kernel for the
antialiased resize (`_upsample_bilinear2d_aa` / `_upsample_bicubic2d_aa`: "compute_index_ranges_weights" not implemented for 'Half'),
so `install_half_stub()` routes `F.interpolate(half tensor, antialias=True)` through fp32 and rounds the result back to fp16 - what the
GPU kernel of the same op does (its accumulation type for Half is float, one rounding on store).  Every other op of the half model runs
ATen's own Half kernels unmodified.
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


_half_stub_installed = False


def install_half_stub():
    """`model.half()` on CPU: the ONE op without a Half kernel is the antialiased resize (modules.py:121; v1.py:274,279).  Compute it in
    fp32 and round once to fp16 - the arithmetic of the GPU kernel (accscalar_t = float for Half).  Everything else stays ATen's."""
    global _half_stub_installed
    if _half_stub_installed:
        return
    import torch.nn.functional as F
    orig = F.interpolate

    def interpolate(input, *a, **k):
        if input.dtype == torch.float16 and k.get("antialias"):
            return orig(input.float(), *a, **k).half()
        return orig(input, *a, **k)

    F.interpolate = interpolate          # moge.model.modules / v1 call `F.interpolate` through the module object
    _half_stub_installed = True


def case_config(case: dict) -> dict:
    """model_config of a case: the named config, optionally with `remap_output` replaced (v2.py:122-136) and heads removed
    (v2.py:46-56: every head is optional; infer() then returns the remaining keys, v2.py:251-298)."""
    import copy
    cfg = copy.deepcopy(oracle_module(case).named_configs()[case["config"]])
    ov = case.get("cfg_override") or {}
    if "remap_output" in ov:
        cfg["remap_output"] = ov["remap_output"]
    for h in ov.get("drop", []):
        cfg.pop(h, None)
    if "head_dim_out" in ov:                       # output convs at levels 0 ... 3 of every head (computed and dropped by v2.py:166)
        for h in ("points_head", "normal_head", "mask_head"):
            if h in cfg:
                cfg[h]["dim_out"] = list(ov["head_dim_out"]) + [cfg[h]["dim_out"][4]]
    return cfg


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
    # BASELINE configs[3] at its own size: moge-2-vitl-normal (all three heads) on 518x518, default tokens (60x60 grid)
    dict(name="vitl_normal_518_t3600", config="moge-2-vitl-normal", seed=0, sane=True, input="rand", input_seed=0, shape=[1, 3, 518, 518],
         kwargs=dict(use_fp16=False), stride=7),
    # the other remap_output modes (v2.py:122-136; every released model uses 'exp').  fov_x is given: the synthetic point head is pinhole-like only
    # under 'exp' (xy * z needs the product); with the other remaps z is nearly constant and the FREE-focal solve is a one-parameter family
    # (reference vs oracle differ by 0.3 there, like tiny_illposed) - with the focal known the shift is well determined; z_bias keeps the un-exponentiated depth away from 0
    dict(name="tiny_remap_linear", config="tiny-vits-normal", cfg_override=dict(remap_output="linear"), z_bias=2.0, seed=0, sane=True, input_seed=21, shape=[2, 3, 98, 126],
         kwargs=dict(num_tokens=120, use_fp16=False, fov_x=50.0)),
    dict(name="tiny_remap_sinh", config="tiny-vits-normal", cfg_override=dict(remap_output="sinh"), z_bias=1.5, seed=0, sane=True, input_seed=22, shape=[2, 3, 84, 112],
         kwargs=dict(num_tokens=108, use_fp16=False, fov_x=65.0)),
    dict(name="tiny_remap_sinh_exp", config="tiny-vits-normal", cfg_override=dict(remap_output="sinh_exp"), seed=0, sane=True, input_seed=23, shape=[1, 3, 140, 150],
         kwargs=dict(num_tokens=56, use_fp16=False, fov_x=40.0)),
    # optional heads (v2.py:46-56, 251-298): no points head -> infer() returns mask (no depth > 0 term) and the masked normal only;
    # points only -> no mask, no metric scale, no normal
    dict(name="tiny_no_points_head", config="tiny-vits-normal", cfg_override=dict(drop=["points_head"]), seed=0, sane=True, input_seed=24, shape=[2, 3, 84, 112],
         kwargs=dict(num_tokens=108, use_fp16=False)),
    # every ConvStack option the released models do not use (modules.py:139-181: pixel_shuffle / nearest / bilinear / conv_transpose at every
    # level of both stacks; modules.py:47-60: GroupNorm(1, C) and GroupNorm(C / 32, C) residual blocks)
    dict(name="tiny_generic_stack", config="tiny-generic-stack", seed=0, sane=True, input_seed=26, shape=[2, 3, 84, 112],
         kwargs=dict(num_tokens=108, use_fp16=False)),
    dict(name="tiny_generic_stack_b", config="tiny-generic-stack-b", seed=1, sane=True, input_seed=27, shape=[1, 3, 98, 126],
         kwargs=dict(num_tokens=120, use_fp16=False)),
    # heads that declare output convs below the last level (modules.py:234-237): the reference computes those maps and drops them (v2.py:166)
    dict(name="tiny_head_side_outputs", config="tiny-vits-normal", cfg_override=dict(head_dim_out=[8, None, 4, 2]), seed=4, sane=True, input_seed=30, shape=[1, 3, 84, 112],
         kwargs=dict(num_tokens=108, use_fp16=False)),
    # ... and every residual-block option (modules.py:31-58, 199-203): SiLU / ELU / LeakyReLU, InstanceNorm2d, hidden width 2x and 4x the level's
    dict(name="tiny_block_options", config="tiny-block-options", seed=2, sane=True, input_seed=28, shape=[2, 3, 84, 112],
         kwargs=dict(num_tokens=108, use_fp16=False)),
    dict(name="tiny_block_options_b", config="tiny-block-options-b", seed=3, sane=True, input_seed=29, shape=[1, 3, 98, 126],
         kwargs=dict(num_tokens=120, use_fp16=False)),
    dict(name="tiny_points_head_only", config="tiny-vits-normal", cfg_override=dict(drop=["mask_head", "normal_head", "scale_head"]), seed=0, sane=True, input_seed=25,
         shape=[2, 3, 84, 112], kwargs=dict(num_tokens=108, use_fp16=False)),
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
    # Head options beyond the class defaults (v1.py:69-71): dim_times_res_block_hidden 2 with two blocks per stage = the layout of the reference's own
    # training recipe (configs/train/v1.json:27-36); 4x with res_block_norm = layer_norm; and that recipe at full size
    dict(name="v1_tiny_hidden_x2", version="v1", config="tiny-v1-vits-x2", seed=2, sane=True, input_seed=14, shape=[2, 3, 98, 126],
         kwargs=dict(num_tokens=120, use_fp16=False)),
    dict(name="v1_tiny_hidden_x4_layer_norm", version="v1", config="tiny-v1-vits-x4-layer", seed=3, sane=True, input_seed=15, shape=[1, 3, 84, 112],
         kwargs=dict(num_tokens=100, use_fp16=False)),
    # output-block options (v1.py:103-109): residual blocks before the last ReLU, a 3x3 last conv - together, and each alone.  Checkpoint seeds chosen
    # for a WELL-CONDITIONED focal / shift solve: with most seeds of these tiny configs 3e-4 of multiplicative noise on the raw point map moves the
    # p99.9 point error by 2-5x from one noise draw to the next (oracle experiment, round 5), which makes an fp16 band a coin toss; seed 11 repeats to 10 %
    dict(name="v1_tiny_last_blocks_conv3", version="v1", config="tiny-v1-vits-last", seed=11, sane=True, input_seed=16, shape=[2, 3, 98, 126],
         kwargs=dict(num_tokens=120, use_fp16=False)),
    dict(name="v1_tiny_last_block_c64", version="v1", config="tiny-v1-vits-last-b", seed=5, sane=True, input_seed=17, shape=[1, 3, 84, 112],
         kwargs=dict(num_tokens=100, use_fp16=False)),
    dict(name="v1_tiny_last_conv3", version="v1", config="tiny-v1-vits-last-c", seed=11, sane=True, input_seed=18, shape=[1, 3, 70, 140],
         kwargs=dict(num_tokens=90, use_fp16=False)),
    dict(name="v1_vitl_train_config_518", version="v1", config="moge-vitl-train-config", seed=0, sane=True, input="rand", input_seed=0, shape=[1, 3, 518, 518],
         kwargs=dict(use_fp16=False), stride=7),
]


def oracle_module(case: dict):
    return O1 if case.get("version") == "v1" else O


def case_state_dict(case: dict, cfg: dict):
    """The synthetic checkpoint of a case (every consumer - reference run, oracle, HIP tests - builds it through here)."""
    sd = oracle_module(case).synth_state_dict(cfg, case["seed"], case["sane"])
    if case.get("massive"):
        O.add_massive_activations(sd, cfg)
    if case.get("z_bias") is not None:              # raw z of the point head = z_bias + O(0.3) noise: keeps a 'linear' / 'sinh' depth away from 0
        sd["points_head.output_blocks.4.bias"][2] = float(case["z_bias"])
    return sd


# the cases whose reference run takes more than a few seconds on 8 cores (the CPU suite replays the oracle on the fast ones only)
SLOW_CASES = ("vits_house518", "vitb_normal_518_t3600", "vitl_518_t3600", "vitl_normal_518x1036", "vitl_normal_1036x518", "v1_vitl_518", "v1_vitl_train_config_518", "vitl_518_t3600_massive",
              "vitl_normal_518_t3600")


def run_reference(case: dict, want=("fp32", "autocast", "half")):
    """-> cfg, sd, x, and the reference's outputs: fp32 infer, fp32 forward, autocast-fp16 infer, .half() infer (None when not in `want`)."""
    install_stubs()
    from moge.model import import_model_class_by_version
    OM = oracle_module(case)
    MoGeModel = import_model_class_by_version(case.get("version", "v2"))
    cfg = case_config(case)
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
    # The reference's OWN fp16 paths on the same input.  (a) fp32 weights + use_fp16=True = torch.autocast(float16) (v2.py:241; what
    # baselines/moge.py and scripts/infer.py run when the model is not .half()): runs on CPU unmodified.  (b) model.half()
    # (scripts/infer.py:83-84 `--fp16`, scripts/app.py:54-56): every tensor fp16 incl. the residual stream; needs install_half_stub().
    kw16 = dict(case["kwargs"]); kw16["use_fp16"] = True
    out16 = model.infer(x, **kw16) if "autocast" in want else None
    out16h = None
    if "half" in want:
        install_half_stub()
        model.half()
        out16h = {k: (v.float() if v.is_floating_point() else v) for k, v in model.infer(x, **kw16).items()}
        model.float()
    return cfg, sd, x, out, fwd, out16, out16h


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


def _strided(k: str, a: np.ndarray, st: int) -> np.ndarray:
    if k in ("intrinsics", "metric_scale") or st <= 1:
        return a
    return a[..., ::st, ::st, :] if (a.ndim >= 3 and a.shape[-1] == 3 and k in ("points", "normal")) else a[..., ::st, ::st]


def drift_stats(a16: dict, ref: dict) -> dict:
    """fp16-vs-fp32 drift of the reference itself in the per-pixel metric the parity tests use (full resolution)."""
    return {k: (dict(flips=MX.mask_flips(a16[k], ref[k])) if ref[k].dtype == torch.bool else MX.summarize(k, a16[k], ref[k])) for k in ref}


def add_half(case: dict) -> None:
    """Append the `.half()` reference outputs (`infer16half.*`, meta.drift16half) to an EXISTING fixture without touching its other arrays;
    the fp32 outputs of this run must reproduce the stored ones (same torch build, same thread count)."""
    path = os.path.join(GOLDEN_DIR, case["name"] + ".npz")
    z = np.load(path)
    blob = {k: z[k] for k in z.files}
    meta = json.loads(bytes(blob["meta"]).decode())
    cfg, sd, x, ref, _fwd, _o16, ref16h = run_reference(case, want=("fp32", "half"))
    st = case.get("stride", 1)
    worst = 0.0
    for k, v in ref.items():
        a, b = _strided(k, v.numpy(), st), blob["infer." + k]
        if b.dtype == np.bool_:
            assert (a == b).all(), (case["name"], k)
        else:
            fin = np.isfinite(b)
            assert (np.isfinite(a) == fin).all()
            worst = max(worst, float(np.abs(a[fin] - b[fin]).max()) if fin.any() else 0.0)
    assert worst <= (1e-2 if not case["sane"] else 2e-5), f"{case['name']}: fp32 rerun differs from the stored golden by {worst}"
    for k, v in ref16h.items():
        blob["infer16half." + k] = _strided(k, v.numpy(), st)
    meta["drift16half"] = drift_stats(ref16h, ref)
    meta["drift16half_source"] = ("reference model.half().infer(): fp16 weights, fp16 residual stream, ATen Half kernels on CPU; the antialiased resize computed in "
                                  "fp32 and rounded to fp16 (oracle/make_golden.py install_half_stub), vs its own fp32 output, full resolution")
    blob["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **blob)
    print(f"{case['name']}: fp32 rerun max |d| {worst:.1e}; half drift " + " ".join(
        f"{k}:{(v.get('p999', v.get('flips'))):.2e}" for k, v in meta["drift16half"].items()), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true", help="compare oracle vs reference, write nothing")
    ap.add_argument("--only", default=None, help="comma-separated case names")
    ap.add_argument("--missing", action="store_true", help="only the cases that have no fixture file yet")
    ap.add_argument("--add-half", action="store_true", help="append the .half() reference outputs to existing fixtures that lack them")
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
    only = set(args.only.split(",")) if args.only else None
    for case in CASES:
        if only and case["name"] not in only:
            continue
        fixture = os.path.join(GOLDEN_DIR, case["name"] + ".npz")
        if args.missing and os.path.exists(fixture):
            continue
        if args.add_half:
            if os.path.exists(fixture) and "infer16half.points" not in np.load(fixture).files and "infer16half.mask" not in np.load(fixture).files:
                add_half(case)
            continue
        cfg, sd, x, ref, ref_fwd, ref16, ref16h = run_reference(case)
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
        if "focal" in tr:
            line.append(f"focal={tr['focal'].tolist()} shift={tr['shift'].tolist()}")
        # drift of the reference's own fp16 paths against its fp32 path, in the per-pixel metric the parity tests use
        drift16, drift16h = drift_stats(ref16, ref), drift_stats(ref16h, ref)
        line.append("ref-fp16 drift: " + " ".join(f"{k}:{(v.get('p999', v.get('flips'))):.2e}" for k, v in drift16.items()))
        line.append("ref-half drift: " + " ".join(f"{k}:{(v.get('p999', v.get('flips'))):.2e}" for k, v in drift16h.items()))
        print("  ".join(line), flush=True)
        if args.check_only:
            continue
        st = case.get("stride", 1)
        blob = {}
        for k, v in ref.items():
            blob["infer." + k] = _strided(k, v.numpy(), st)
        for k, v in ref16.items():          # the reference's fp16 (autocast) outputs
            blob["infer16." + k] = _strided(k, v.numpy(), st)
        for k, v in ref16h.items():         # the reference's .half() outputs
            blob["infer16half." + k] = _strided(k, v.numpy(), st)
        for k, v in ref_fwd.items():
            blob["forward." + k] = _strided(k, v.detach().numpy(), st)
        meta = dict(case=case, weights_sha256=weights_digest(sd), torch=torch.__version__, scipy=scipy.__version__,
                    numpy=np.__version__, threads=torch.get_num_threads(),
                    input_sha256=hashlib.sha256(x.numpy().tobytes()).hexdigest(),
                    focal=tr["focal"].tolist() if "focal" in tr else None, shift=tr["shift"].tolist() if "shift" in tr else None, drift16=drift16,
                    drift16_source="reference infer(use_fp16=True): fp32 weights under torch.autocast(cpu, float16), vs its own use_fp16=False output, full resolution",
                    drift16half=drift16h,
                    drift16half_source="reference model.half().infer(): fp16 weights, fp16 residual stream, ATen Half kernels on CPU; the antialiased resize computed in "
                                       "fp32 and rounded to fp16 (install_half_stub), vs its own fp32 output, full resolution")
        blob["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(fixture, **blob)


if __name__ == "__main__":
    main()
