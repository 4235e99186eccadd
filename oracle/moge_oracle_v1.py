"""
CPU oracle for the MoGe-1 `moge.model.v1.MoGeModel.infer()` path.  TEST INFRASTRUCTURE ONLY - same rules as
oracle/moge_oracle.py: only tests/, smoke() and the cpu_baseline leg of bench.py may import it.

Functional restatement (torch CPU ops, fp32) of, paths relative to /root/reference:
  ResidualConvBlock   moge/model/v1.py:24-58   (GroupNorm(1, C) -> ReLU -> 3x3 -> GroupNorm(C/32, C) -> ReLU -> 3x3, + identity skip)
  Head                moge/model/v1.py:61-142  (sum of 1x1 projections of the ViT taps; three [uv concat, ConvTranspose2d k2 s2, 3x3 replicate,
                                                res blocks] stages; bilinear resize to the resized image; uv concat; per-output 3x3 -> ReLU -> 1x1)
  MoGeModel.forward   moge/model/v1.py:269-300 (bicubic antialiased resize to ~num_tokens*196 pixels, normalise, bilinear antialiased resize to
                                                multiples of 14, ViT taps = the LAST n blocks with the final norm, head, bilinear resize back, remap)
  MoGeModel.infer     moge/model/v1.py:302-391 (mask = raw head output > mask_threshold; focal / shift recovery; intrinsics; re-projection; masking -
                                                no `depth > 0` term, no metric scale, no normals)
Pinned like the v2 oracle: oracle/make_golden.py runs the real v1 class on synthetic checkpoints and commits the outputs (tests/golden/v1_*.npz).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import moge_oracle as O2

VIT_SPECS = O2.VIT_SPECS
PATCH = O2.PATCH


def make_config(encoder: str = "dinov2_vitl14", intermediate_layers=4, dim_proj: int = 512, dim_upsample=(256, 128, 128), num_res_blocks: int = 1,
                remap_output: str = "exp", num_tokens_range=(1200, 2500), last_conv_channels: int = 32, mask_threshold: float = 0.5,
                dim_times_res_block_hidden: int = 1, res_block_norm: str = "group_norm", last_res_blocks: int = 0, last_conv_size: int = 1) -> dict:
    """`model_config` of a MoGe-1 checkpoint (v1.py:148-163).  The defaults are the class's; what Ruicheng/moge-vitl carries lives in the
    (unreachable) HF checkpoint - the repo's own training recipe configs/train/v1.json:27-36 uses dim_upsample [256, 128, 64],
    dim_times_res_block_hidden 2, num_res_blocks 2 ("moge-vitl-train-config" below)."""
    return dict(encoder=encoder, intermediate_layers=intermediate_layers, dim_proj=dim_proj, dim_upsample=list(dim_upsample),
                dim_times_res_block_hidden=dim_times_res_block_hidden, num_res_blocks=num_res_blocks, remap_output=remap_output, res_block_norm=res_block_norm,
                num_tokens_range=list(num_tokens_range), last_res_blocks=last_res_blocks, last_conv_channels=last_conv_channels, last_conv_size=last_conv_size,
                mask_threshold=mask_threshold)


def named_configs() -> Dict[str, dict]:
    return {
        "moge-vitl": make_config(),
        "tiny-v1-vits": make_config("dinov2_vits14", 4, 128, (64, 64, 32), 1, "exp", (60, 200)),
        # configs/train/v1.json:27-36 (the reference's own MoGe-1 recipe): hidden width 2x, two residual blocks per stage
        "moge-vitl-train-config": make_config("dinov2_vitl14", 4, 512, (256, 128, 64), 2, "exp", (1200, 2500), dim_times_res_block_hidden=2),
        "tiny-v1-vits-x2": make_config("dinov2_vits14", 4, 128, (64, 64, 32), 2, "exp", (60, 200), dim_times_res_block_hidden=2),
        "tiny-v1-vits-x4-layer": make_config("dinov2_vits14", 4, 128, (64, 32, 32), 1, "exp", (60, 200), dim_times_res_block_hidden=4, res_block_norm="layer_norm"),
        # output-block options (v1.py:103-109): residual blocks on the last_conv_channels map, a 3x3 last conv
        "tiny-v1-vits-last": make_config("dinov2_vits14", 4, 128, (64, 64, 32), 1, "exp", (60, 200), dim_times_res_block_hidden=2, last_res_blocks=2, last_conv_size=3),
        "tiny-v1-vits-last-b": make_config("dinov2_vits14", 4, 128, (64, 32, 32), 1, "exp", (60, 200), last_conv_channels=64, res_block_norm="layer_norm", last_res_blocks=1),
        "tiny-v1-vits-last-c": make_config("dinov2_vits14", 4, 128, (64, 32, 32), 1, "exp", (60, 200), last_conv_size=3),
    }


def tap_layers(cfg: dict) -> List[int]:
    D, L, _ = VIT_SPECS[cfg["encoder"]]
    n = cfg["intermediate_layers"]
    return list(range(L - n, L)) if isinstance(n, int) else list(n)        # vision_transformer.py:286 (int n = the last n blocks)


def state_dict_spec(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    D, L, _ = VIT_SPECS[cfg["encoder"]]
    bb = "backbone."
    out: List[Tuple[str, Tuple[int, ...]]] = [
        (bb + "cls_token", (1, 1, D)), (bb + "pos_embed", (1, 1 + O2.POS_GRID ** 2, D)), (bb + "mask_token", (1, D)),
        (bb + "patch_embed.proj.weight", (D, 3, PATCH, PATCH)), (bb + "patch_embed.proj.bias", (D,))]
    for i in range(L):
        p = f"{bb}blocks.{i}."
        out += [(p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)), (p + "attn.qkv.weight", (3 * D, D)), (p + "attn.qkv.bias", (3 * D,)),
                (p + "attn.proj.weight", (D, D)), (p + "attn.proj.bias", (D,)), (p + "ls1.gamma", (D,)),
                (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,)), (p + "mlp.fc1.weight", (4 * D, D)), (p + "mlp.fc1.bias", (4 * D,)),
                (p + "mlp.fc2.weight", (D, 4 * D)), (p + "mlp.fc2.bias", (D,)), (p + "ls2.gamma", (D,))]
    out += [(bb + "norm.weight", (D,)), (bb + "norm.bias", (D,))]
    P, ups, c4 = cfg["dim_proj"], cfg["dim_upsample"], cfg["last_conv_channels"]
    for k in range(len(tap_layers(cfg))):
        out += [(f"head.projects.{k}.weight", (P, D, 1, 1)), (f"head.projects.{k}.bias", (P,))]
    for i, (ci, co) in enumerate(zip([P] + ups[:-1], ups)):
        u = f"head.upsample_blocks.{i}."
        out += [(u + "0.0.weight", (ci + 2, co, 2, 2)), (u + "0.0.bias", (co,)), (u + "0.1.weight", (co, co, 3, 3)), (u + "0.1.bias", (co,))]
        ch = co * cfg.get("dim_times_res_block_hidden", 1)          # hidden width (v1.py:85)
        for j in range(cfg["num_res_blocks"]):
            r = f"{u}{1 + j}.layers."
            out += [(r + "0.weight", (co,)), (r + "0.bias", (co,)), (r + "2.weight", (ch, co, 3, 3)), (r + "2.bias", (ch,)),
                    (r + "3.weight", (ch,)), (r + "3.bias", (ch,)), (r + "5.weight", (co, ch, 3, 3)), (r + "5.bias", (co,))]
    nl, ks, ch4 = cfg.get("last_res_blocks", 0), cfg.get("last_conv_size", 1), c4 * cfg.get("dim_times_res_block_hidden", 1)
    for o, dim_out in enumerate((3, 1)):
        # nn.Sequential(conv3x3, *ResidualConvBlock x last_res_blocks, ReLU, Conv2d(k = last_conv_size))   (v1.py:103-109)
        b = f"head.output_block.{o}."
        out += [(b + "0.weight", (c4, ups[-1] + 2, 3, 3)), (b + "0.bias", (c4,))]
        for j in range(nl):
            r = f"{b}{1 + j}.layers."
            out += [(r + "0.weight", (c4,)), (r + "0.bias", (c4,)), (r + "2.weight", (ch4, c4, 3, 3)), (r + "2.bias", (ch4,)),
                    (r + "3.weight", (ch4,)), (r + "3.bias", (ch4,)), (r + "5.weight", (c4, ch4, 3, 3)), (r + "5.bias", (c4,))]
        out += [(b + f"{nl + 2}.weight", (dim_out, c4, ks, ks)), (b + f"{nl + 2}.bias", (dim_out,))]
    out += [("image_mean", (1, 3, 1, 1)), ("image_std", (1, 3, 1, 1))]
    return out


def synth_state_dict(cfg: dict, seed: int = 0, sane_geometry: bool = True) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights, O(1) activations; with `sane_geometry` the points output block reads the uv channels of its input so
    the raw point map is pinhole-like and the focal / shift solve is well posed; the mask output straddles the 0.5 threshold."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in state_dict_spec(cfg):
        leaf = key.rsplit(".", 1)[-1]
        if key == "image_mean":
            t = torch.tensor(O2.IMAGE_MEAN).view(1, 3, 1, 1)
        elif key == "image_std":
            t = torch.tensor(O2.IMAGE_STD).view(1, 3, 1, 1)
        elif key.endswith("pos_embed"):
            t = 0.2 * torch.randn(shape, generator=g)
        elif key.endswith("cls_token") or key.endswith("mask_token"):
            t = 0.5 * torch.randn(shape, generator=g)
        elif len(shape) == 1 and leaf == "weight":          # LayerNorm / GroupNorm gains
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif leaf == "gamma":
            t = 0.05 + 0.25 * torch.rand(shape, generator=g)
        elif leaf == "bias":
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[0] if key.endswith(".0.0.weight") else int(np.prod(shape[1:]))      # ConvTranspose2d: (Cin, Cout, 2, 2)
            gain = 0.5 if ".layers.5." in key else (1.4 if ".layers.2." in key else 1.0)
            t = gain * torch.randn(shape, generator=g) / math.sqrt(fan_in)
        sd[key] = t.float().contiguous()
    c_last = cfg["dim_upsample"][-1]
    last, kc = cfg.get("last_res_blocks", 0) + 2, cfg.get("last_conv_size", 1) // 2          # index of the last conv in its Sequential, its centre tap
    if sane_geometry:
        w0 = sd["head.output_block.0.0.weight"]             # (c4, C + 2, 3, 3): channels C, C+1 are (u, v)
        w0.mul_(0.3)
        w0[0, c_last, 1, 1] = 4.0
        w0[1, c_last + 1, 1, 1] = 4.0
        w0[2, c_last, 1, 1] = -4.0                          # relu(+u), relu(-u): both signs survive the ReLU
        w0[3, c_last + 1, 1, 1] = -4.0
        w2 = sd[f"head.output_block.0.{last}.weight"]
        w2.mul_(0.05)
        w2[0, 0, kc, kc], w2[0, 2, kc, kc] = 0.3, -0.3
        w2[1, 1, kc, kc], w2[1, 3, kc, kc] = 0.3, -0.3
        sd[f"head.output_block.0.{last}.bias"].copy_(torch.tensor([0.0, 0.0, 0.4]))
    sd[f"head.output_block.1.{last}.weight"].mul_(3.0)
    sd[f"head.output_block.1.{last}.bias"].fill_(0.9)
    return sd


def save_checkpoint(path: str, cfg: dict, sd: Dict[str, torch.Tensor]) -> None:
    torch.save({"model_config": cfg, "model": sd}, path)


def resized_dims(H: int, W: int, num_tokens: int) -> Tuple[int, int]:
    """(resized_height, resized_width) of v1.py:272-274: python float arithmetic, int() truncation."""
    f = ((num_tokens * 14 ** 2) / (H * W)) ** 0.5
    return int(H * f), int(W * f)


def res_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str, norm: str = "group_norm") -> torch.Tensor:
    """v1.py:44-58: GroupNorm(1, C) -> ReLU -> 3x3 (C -> Ch) -> GroupNorm(Ch // 32 | 1, Ch) -> ReLU -> 3x3 (Ch -> C), + x; Ch = the conv's own width"""
    h = F.relu(F.group_norm(x, 1, sd[p + "0.weight"], sd[p + "0.bias"]))
    h = F.conv2d(F.pad(h, (1, 1, 1, 1), mode="replicate"), sd[p + "2.weight"], sd[p + "2.bias"])
    h = F.relu(F.group_norm(h, h.shape[1] // 32 if norm == "group_norm" else 1, sd[p + "3.weight"], sd[p + "3.bias"]))
    h = F.conv2d(F.pad(h, (1, 1, 1, 1), mode="replicate"), sd[p + "5.weight"], sd[p + "5.bias"])
    return h + x


def with_uv(x: torch.Tensor, aspect: float) -> torch.Tensor:
    uv = O2.view_plane_uv(x.shape[-1], x.shape[-2], aspect, dtype=x.dtype).permute(2, 0, 1)[None].expand(x.shape[0], -1, -1, -1)
    return torch.cat([x, uv], dim=1)


def forward(cfg: dict, sd: Dict[str, torch.Tensor], image: torch.Tensor, num_tokens: int, trace: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    B, _, H, W = image.shape
    D, L, heads = VIT_SPECS[cfg["encoder"]]
    rh, rw = resized_dims(H, W, num_tokens)
    img = F.interpolate(image, (rh, rw), mode="bicubic", align_corners=False, antialias=True)
    img = (img - sd["image_mean"]) / sd["image_std"]
    ph, pw = rh // 14, rw // 14
    x14 = F.interpolate(img, (ph * 14, pw * 14), mode="bilinear", align_corners=False, antialias=True)
    if trace is not None:
        trace["image_14"] = x14
    bb = "backbone."
    x = F.conv2d(x14, sd[bb + "patch_embed.proj.weight"], sd[bb + "patch_embed.proj.bias"], stride=PATCH).flatten(2).transpose(1, 2)
    x = torch.cat([sd[bb + "cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + O2.pos_embed_for_grid(sd[bb + "pos_embed"], ph, pw)
    taps, outs = tap_layers(cfg), []
    for i in range(L):
        x = O2.vit_block(x, sd, f"{bb}blocks.{i}.", heads)
        if i in taps:
            outs.append(F.layer_norm(x, (D,), sd[bb + "norm.weight"], sd[bb + "norm.bias"], 1e-6))
    feat = None
    for k, o in enumerate(outs):
        f = F.conv2d(o[:, 1:].permute(0, 2, 1).reshape(B, D, ph, pw), sd[f"head.projects.{k}.weight"], sd[f"head.projects.{k}.bias"])
        feat = f if feat is None else feat + f
    if trace is not None:
        trace["proj"] = feat
    aspect = rw / rh                                          # Head.forward uses the RESIZED image's aspect ratio (v1.py:118)
    x = feat
    for i in range(len(cfg["dim_upsample"])):
        u = f"head.upsample_blocks.{i}."
        x = with_uv(x, aspect)
        x = F.conv_transpose2d(x, sd[u + "0.0.weight"], sd[u + "0.0.bias"], stride=2)
        x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), sd[u + "0.1.weight"], sd[u + "0.1.bias"])
        for j in range(cfg["num_res_blocks"]):
            x = res_block(x, sd, f"{u}{1 + j}.layers.", cfg.get("res_block_norm", "group_norm"))
        if trace is not None:
            trace[f"up{i}"] = x
    x = F.interpolate(x, (rh, rw), mode="bilinear", align_corners=False)
    x = with_uv(x, aspect)
    outs2 = []
    for o in range(2):
        b = f"head.output_block.{o}."
        y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), sd[b + "0.weight"], sd[b + "0.bias"])
        nl, ks = cfg.get("last_res_blocks", 0), cfg.get("last_conv_size", 1)
        for j in range(nl):                                   # v1.py:106
            y = res_block(y, sd, f"{b}{1 + j}.layers.", cfg.get("res_block_norm", "group_norm"))
        y = F.relu(y)
        if ks > 1:
            y = F.pad(y, (ks // 2,) * 4, mode="replicate")
        outs2.append(F.conv2d(y, sd[b + f"{nl + 2}.weight"], sd[b + f"{nl + 2}.bias"]))
    points = F.interpolate(outs2[0], (H, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    mask = F.interpolate(outs2[1], (H, W), mode="bilinear", align_corners=False).squeeze(1)
    return {"points": O2.remap_points(points, cfg["remap_output"]), "mask": mask}


def infer(cfg: dict, sd: Dict[str, torch.Tensor], image: torch.Tensor, fov_x=None, resolution_level: int = 9, num_tokens: Optional[int] = None,
          apply_mask: bool = True, force_projection: bool = True, trace: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    squeeze = image.dim() == 3
    if squeeze:
        image = image[None]
    image = image.float()
    H, W = image.shape[-2:]
    aspect = W / H
    if num_tokens is None:
        lo, hi = cfg["num_tokens_range"]
        num_tokens = int(lo + (resolution_level / 9) * (hi - lo))
    out = forward(cfg, sd, image, num_tokens, trace)
    if trace is not None:
        trace["forward"] = {k: v.clone() for k, v in out.items()}
    points, mask = out["points"], out["mask"]
    mask_b = mask > cfg["mask_threshold"]
    if fov_x is None:
        focal, shift = O2.recover_focal_shift(points, mask_b)
    else:
        fov = torch.as_tensor(fov_x, dtype=points.dtype)
        focal = aspect / (1 + aspect ** 2) ** 0.5 / torch.tan(torch.deg2rad(fov / 2))
        if focal.ndim == 0:
            focal = focal[None].expand(points.shape[0])
        _, shift = O2.recover_focal_shift(points, mask_b, focal)
    if trace is not None:
        trace["focal"], trace["shift"] = focal.clone(), shift.clone()
    fx = focal / 2 * (1 + aspect ** 2) ** 0.5 / aspect
    fy = focal / 2 * (1 + aspect ** 2) ** 0.5
    K = O2._intrinsics(fx, fy)
    depth = points[..., 2] + shift[..., None, None]
    if force_projection:
        points = O2._depth_to_points(depth, K)
    else:
        points = points + torch.stack([torch.zeros_like(shift), torch.zeros_like(shift), shift], dim=-1)[..., None, None, :]
    if apply_mask:
        points = torch.where(mask_b[..., None], points, torch.inf)
        depth = torch.where(mask_b, depth, torch.inf)
    res = {"points": points, "intrinsics": K, "depth": depth, "mask": mask_b}
    return {k: v.squeeze(0) for k, v in res.items()} if squeeze else res
