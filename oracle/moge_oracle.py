"""
CPU oracle for the MoGe-2 `MoGeModel.infer()` hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional, single-file restatement (torch CPU ops, fp32 by default, fp64 on request)
of the reference algorithm.  It is *not* part of the product: only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it.  The product path (`moge_amd`) never does.

Parity pinning: the reference ships no tests / golden vectors for this path, so
this oracle is pinned against *outputs of the reference itself*, run in the build container by
`oracle/make_golden.py` (imports /root/reference with stubs for the un-installed `utils3d`/`cv2`).
The fixtures it wrote live in `tests/golden/`; `tests/test_oracle_golden.py` re-checks the oracle
against them on every run.  The `utils3d` boundary (two calls) is "parity unpinned": the package is
not vendored in the reference, semantics are restated from the call sites (see `_intrinsics`,
`_depth_to_points`).

Reference sections followed (paths relative to /root/reference):
  forward            moge/model/v2.py:138-192
  infer              moge/model/v2.py:194-303
  encoder wrapper    moge/model/modules.py:120-136
  ViT                moge/model/dinov2/models/vision_transformer.py:187-243,283-333
  block/attn/mlp     moge/model/dinov2/layers/{block.py:110-112,attention.py:70-81,mlp.py:34-40,layer_scale.py:27}
  conv stack         moge/model/modules.py:18-68,139-182,195-254
  uv grid            moge/utils/geometry_torch.py:40-52
  focal/shift        moge/utils/geometry_torch.py:115-170, moge/utils/geometry_numpy.py:79-112
  LM solver          scipy.optimize.least_squares(method='lm') -> MINPACK lmdif (restated in oracle/lmdif.py)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .lmdif import lmdif_scalar

# (embed_dim, depth, heads) of the DINOv2 backbones MoGe uses (vision_transformer.py:351-390)
VIT_SPECS = {
    "dinov2_vits14": (384, 12, 6),
    "dinov2_vitb14": (768, 12, 12),
    "dinov2_vitl14": (1024, 24, 16),
}
PATCH = 14
POS_GRID = 37  # 518 / 14, the pre-training grid of pos_embed (backbones.py:21)
IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)
HEAD_NAMES = ("points_head", "normal_head", "mask_head")


# --------------------------------------------------------------------------------------------
# configs + synthetic checkpoints (reference .pt format: {'model_config':..., 'model': state_dict})
# --------------------------------------------------------------------------------------------
def make_config(backbone: str = "dinov2_vitl14", taps: Sequence[int] = (5, 11, 17, 23),
                dims: Sequence[int] = (1024, 256, 128, 64, 32), normal: bool = True,
                scale_hidden: Optional[int] = None, num_tokens_range=(1200, 3600),
                neck_resamplers=None, head_resamplers=None, neck_norms=("none", "none"), head_norms=("none", "none"),
                neck_block=("relu", 1), head_block=("relu", 1)) -> dict:
    """model_config in the shape of configs/train/v2.json:238-285 (vitl) for any backbone/dims; the ConvStack options the released models
    leave at [conv_transpose x3, bilinear] / no norms / ReLU / hidden = width (modules.py:139-181, 31-60, 199-203) can be set for the
    generic-layout test configs: *_block = (activation, dim_times_res_block_hidden), written to the config only when not the default."""
    D = VIT_SPECS[backbone][0]
    dims = list(dims)
    resamplers = ["conv_transpose", "conv_transpose", "conv_transpose", "bilinear"]

    def block_opts(blk):
        out = {}
        if blk[0] != "relu":
            out["activation"] = blk[0]
        if blk[1] != 1:
            out["dim_times_res_block_hidden"] = blk[1]
        return out

    def head(cout):
        return {"dim_in": list(dims), "dim_out": [None, None, None, None, cout], "dim_res_blocks": list(dims),
                "num_res_blocks": [0, 1, 1, 1, 0], "res_block_in_norm": head_norms[0], "res_block_hidden_norm": head_norms[1],
                "resamplers": list(head_resamplers or resamplers), **block_opts(head_block)}

    cfg = {
        "encoder": {"backbone": backbone, "intermediate_layers": list(taps), "dim_out": dims[0]},
        "neck": {"dim_in": [dims[0] + 2, 2, 2, 2, 2], "dim_out": None, "dim_res_blocks": list(dims),
                 "num_res_blocks": [0, 2, 2, 2, 0], "res_block_in_norm": neck_norms[0], "res_block_hidden_norm": neck_norms[1],
                 "resamplers": list(neck_resamplers or resamplers), **block_opts(neck_block)},
        "points_head": head(3),
        "mask_head": head(1),
        "scale_head": {"dims": [D, scale_hidden or D, scale_hidden or D, 1]},
        "remap_output": "exp",
        "num_tokens_range": list(num_tokens_range),
    }
    if normal:
        cfg["normal_head"] = head(3)
    return cfg


def named_configs() -> Dict[str, dict]:
    """vitl(-normal) is the repo's config; vitb/vits decoder dims are NOT in the reference repo (they ship inside the
    HF checkpoints) - the values here are the param-count-matching guesses of SURVEY.md 8(c); 'tiny' is a test size."""
    return {
        "moge-2-vitl-normal": make_config("dinov2_vitl14", (5, 11, 17, 23), (1024, 256, 128, 64, 32), True),
        "moge-2-vitl": make_config("dinov2_vitl14", (5, 11, 17, 23), (1024, 256, 128, 64, 32), False),
        "moge-2-vitb-normal": make_config("dinov2_vitb14", (2, 5, 8, 11), (512, 256, 128, 64, 32), True),
        "moge-2-vits-normal": make_config("dinov2_vits14", (2, 5, 8, 11), (384, 256, 128, 64, 32), True),
        "tiny-vits-normal": make_config("dinov2_vits14", (2, 5, 8, 11), (128, 64, 64, 32, 32), True, scale_hidden=128),
        # every ConvStack option of the decoder that no released model uses (modules.py:139-181, 47-60): all four x2 up-samplers in both
        # stacks, GroupNorm(1, C) / GroupNorm(C / 32, C) residual blocks
        "tiny-generic-stack": make_config("dinov2_vits14", (2, 5, 8, 11), (128, 64, 64, 32, 32), True, scale_hidden=128,
                                          neck_resamplers=["pixel_shuffle", "nearest", "bilinear", "conv_transpose"],
                                          head_resamplers=["nearest", "conv_transpose", "pixel_shuffle", "bilinear"],
                                          neck_norms=("layer_norm", "group_norm"), head_norms=("none", "layer_norm")),
        "tiny-generic-stack-b": make_config("dinov2_vits14", (2, 5, 8, 11), (128, 64, 64, 32, 32), True, scale_hidden=128,
                                            neck_resamplers=["bilinear", "conv_transpose", "pixel_shuffle", "nearest"],
                                            head_resamplers=["conv_transpose", "pixel_shuffle", "bilinear", "conv_transpose"],
                                            neck_norms=("none", "none"), head_norms=("group_norm", "none")),
        # ... and the residual-block options (modules.py:31-58, 199-203): the other activations, InstanceNorm2d, hidden width = k x width
        "tiny-block-options": make_config("dinov2_vits14", (2, 5, 8, 11), (128, 64, 64, 32, 32), True, scale_hidden=128,
                                          neck_norms=("instance_norm", "group_norm"), head_norms=("none", "instance_norm"),
                                          neck_block=("silu", 2), head_block=("elu", 1)),
        "tiny-block-options-b": make_config("dinov2_vits14", (2, 5, 8, 11), (128, 64, 64, 32, 32), True, scale_hidden=128,
                                            neck_resamplers=["nearest", "conv_transpose", "conv_transpose", "pixel_shuffle"],
                                            neck_norms=("none", "none"), head_norms=("layer_norm", "none"),
                                            neck_block=("relu", 2), head_block=("leaky_relu", 4)),
    }


def state_dict_spec(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (key, shape) list of the reference state dict for `cfg`."""
    D, L, _ = VIT_SPECS[cfg["encoder"]["backbone"]]
    out: List[Tuple[str, Tuple[int, ...]]] = []
    bb = "encoder.backbone."
    out += [(bb + "cls_token", (1, 1, D)), (bb + "pos_embed", (1, 1 + POS_GRID * POS_GRID, D)),
            (bb + "mask_token", (1, D)),
            (bb + "patch_embed.proj.weight", (D, 3, PATCH, PATCH)), (bb + "patch_embed.proj.bias", (D,))]
    for i in range(L):
        p = f"{bb}blocks.{i}."
        out += [(p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)),
                (p + "attn.qkv.weight", (3 * D, D)), (p + "attn.qkv.bias", (3 * D,)),
                (p + "attn.proj.weight", (D, D)), (p + "attn.proj.bias", (D,)),
                (p + "ls1.gamma", (D,)),
                (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,)),
                (p + "mlp.fc1.weight", (4 * D, D)), (p + "mlp.fc1.bias", (4 * D,)),
                (p + "mlp.fc2.weight", (D, 4 * D)), (p + "mlp.fc2.bias", (D,)),
                (p + "ls2.gamma", (D,))]
    out += [(bb + "norm.weight", (D,)), (bb + "norm.bias", (D,))]
    c0 = cfg["encoder"]["dim_out"]
    for k in range(len(cfg["encoder"]["intermediate_layers"])):
        out += [(f"encoder.output_projections.{k}.weight", (c0, D, 1, 1)), (f"encoder.output_projections.{k}.bias", (c0,))]
    out += [("encoder.image_mean", (1, 3, 1, 1)), ("encoder.image_std", (1, 3, 1, 1))]

    def stack(name, sc):
        dims = sc["dim_res_blocks"]
        dim_in = sc["dim_in"]
        dim_out = sc["dim_out"] if isinstance(sc["dim_out"], list) else [sc["dim_out"]] * len(dims)
        nres = sc["num_res_blocks"]
        for l, c in enumerate(dims):
            if dim_in[l] is not None:
                out.append((f"{name}.input_blocks.{l}.weight", (c, dim_in[l], 1, 1)))
                out.append((f"{name}.input_blocks.{l}.bias", (c,)))
        for l in range(len(dims) - 1):
            cin, cout = dims[l], dims[l + 1]
            kind = sc["resamplers"][l]
            if kind == "conv_transpose":
                out.append((f"{name}.resamplers.{l}.0.weight", (cin, cout, 2, 2)))
                out.append((f"{name}.resamplers.{l}.0.bias", (cout,)))
                out.append((f"{name}.resamplers.{l}.1.weight", (cout, cout, 3, 3)))
                out.append((f"{name}.resamplers.{l}.1.bias", (cout,)))
            elif kind in ("bilinear", "nearest"):
                out.append((f"{name}.resamplers.{l}.1.weight", (cout, cin, 3, 3)))
                out.append((f"{name}.resamplers.{l}.1.bias", (cout,)))
            elif kind == "pixel_shuffle":           # Conv2d(cin, 4 cout, 3) -> PixelShuffle(2) -> Conv2d(cout, cout, 3)   (modules.py:146-151)
                out.append((f"{name}.resamplers.{l}.0.weight", (4 * cout, cin, 3, 3)))
                out.append((f"{name}.resamplers.{l}.0.bias", (4 * cout,)))
                out.append((f"{name}.resamplers.{l}.2.weight", (cout, cout, 3, 3)))
                out.append((f"{name}.resamplers.{l}.2.bias", (cout,)))
            else:
                raise NotImplementedError(kind)
        mult = sc.get("dim_times_res_block_hidden", 1)
        for l, c in enumerate(dims):
            ch = c * mult                                       # hidden width (modules.py:222)
            for j in range(nres[l]):
                # GroupNorm affine parameters (modules.py:47-50); InstanceNorm2d(C) has none (affine=False, no running statistics)
                # (key order = the order the synthetic weights are drawn in: norms first, as the committed fixtures were made)
                if sc["res_block_in_norm"] not in ("none", "instance_norm"):
                    out.append((f"{name}.res_blocks.{l}.{j}.layers.0.weight", (c,)))
                    out.append((f"{name}.res_blocks.{l}.{j}.layers.0.bias", (c,)))
                if sc["res_block_hidden_norm"] not in ("none", "instance_norm"):
                    out.append((f"{name}.res_blocks.{l}.{j}.layers.3.weight", (ch,)))
                    out.append((f"{name}.res_blocks.{l}.{j}.layers.3.bias", (ch,)))
                out.append((f"{name}.res_blocks.{l}.{j}.layers.2.weight", (ch, c, 3, 3)))
                out.append((f"{name}.res_blocks.{l}.{j}.layers.2.bias", (ch,)))
                out.append((f"{name}.res_blocks.{l}.{j}.layers.5.weight", (c, ch, 3, 3)))
                out.append((f"{name}.res_blocks.{l}.{j}.layers.5.bias", (c,)))
        for l, c in enumerate(dims):
            if dim_out[l] is not None:
                out.append((f"{name}.output_blocks.{l}.weight", (dim_out[l], c, 1, 1)))
                out.append((f"{name}.output_blocks.{l}.bias", (dim_out[l],)))

    stack("neck", cfg["neck"])
    for h in HEAD_NAMES:
        if cfg.get(h) is not None:
            stack(h, cfg[h])
    if cfg.get("scale_head") is not None:
        d = cfg["scale_head"]["dims"]
        for i in range(len(d) - 1):
            out += [(f"scale_head.{2 * i}.weight", (d[i + 1], d[i])), (f"scale_head.{2 * i}.bias", (d[i + 1],))]
    return out


def synth_state_dict(cfg: dict, seed: int = 0, sane_geometry: bool = True) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights (no pretrained checkpoint is reachable offline).

    Scales are chosen so activations stay O(1) through the whole net (so fp16 mode does not overflow and
    relative errors are meaningful).  With `sane_geometry` the last decoder level is biased so that the raw
    point head output is roughly (s*u, s*v, small) - i.e. a pinhole-like point map - which makes the
    focal/shift solve well-posed; the mask head bias/gain is set so the logits straddle 0 with few values
    near 0 (with plain random init the mask is all-False and the recovery is untested).
    `sane_geometry=False` leaves everything random (ill-posed solve; stress case).
    """
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in state_dict_spec(cfg):
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("image_mean"):
            t = torch.tensor(IMAGE_MEAN).view(1, 3, 1, 1)
        elif key.endswith("image_std"):
            t = torch.tensor(IMAGE_STD).view(1, 3, 1, 1)
        elif key.endswith("pos_embed"):
            t = 0.2 * torch.randn(shape, generator=g)
        elif key.endswith("cls_token") or key.endswith("mask_token"):
            t = 0.5 * torch.randn(shape, generator=g)
        elif "norm" in key and leaf == "weight" and len(shape) == 1:
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif leaf == "weight" and len(shape) == 1:          # GroupNorm weights of normalised residual blocks (generic-layout configs only)
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif leaf == "gamma":
            t = 0.05 + 0.25 * torch.rand(shape, generator=g)
        elif leaf == "bias":
            t = 0.1 * torch.randn(shape, generator=g)
        else:  # weights: fan-in scaled
            if "resamplers" in key and key.endswith(".0.weight"):
                fan_in = shape[0]                      # ConvTranspose2d: (Cin, Cout, 2, 2), one tap per output pixel
            else:
                fan_in = int(np.prod(shape[1:]))
            gain = 1.0
            if "res_blocks" in key:
                gain = 1.4 if ".layers.2." in key else 0.45     # ReLU halves the second moment; keep the branch small
            elif "_head.input_blocks" in key:
                gain = 0.6
            t = gain * torch.randn(shape, generator=g) / math.sqrt(fan_in)
        sd[key] = t.float().contiguous()

    if sane_geometry:
        # level-4 uv injection -> channels 0/1 of the neck's level-4 map carry (u, v)
        w = sd["neck.input_blocks.4.weight"]
        w.mul_(0.3)
        w[0, 0, 0, 0] = 4.0
        w[1, 1, 0, 0] = 4.0
        rs = sd["neck.resamplers.3.%s.weight" % ("2" if cfg["neck"]["resamplers"][3] == "pixel_shuffle" else "1")]
        rs[0:2].mul_(0.15)
        ph = "points_head."
    if sane_geometry and cfg.get("points_head") is not None:
        w = sd[ph + "input_blocks.4.weight"]
        w[0:2].mul_(0.1)
        w[0, 0, 0, 0] = 1.0
        w[1, 1, 0, 0] = 1.0
        sd[ph + "resamplers.3.%s.weight" % ("2" if cfg["points_head"]["resamplers"][3] == "pixel_shuffle" else "1")][0:2].mul_(0.1)
        wo = sd[ph + "output_blocks.4.weight"]
        wo.mul_(0.25)
        wo[0].mul_(0.2); wo[1].mul_(0.2)
        wo[0, 0, 0, 0] = 0.35
        wo[1, 1, 0, 0] = 0.35
        sd[ph + "output_blocks.4.bias"].copy_(torch.tensor([0.0, 0.0, 0.4]))
    # mask logits: spread them and move the mean so ~70% of the pixels are valid
    if cfg.get("mask_head") is not None:
        sd["mask_head.output_blocks.4.weight"].mul_(4.0)
        sd["mask_head.output_blocks.4.bias"].fill_(-4.0)
    return sd


def add_massive_activations(sd: Dict[str, torch.Tensor], cfg: dict, block: int = 2,
                            channels=((7, 450.0), (300, -380.0), (901, 600.0)), mean_offset: float = 3.0) -> Dict[str, torch.Tensor]:
    """Put the residual stream into the numerical regime of PRETRAINED DINOv2 weights (SURVEY.md 7 hard-part 3; block.py:110-112): from
    `block` on, a few residual channels sit hundreds of times above the rest ("massive activations") and every token carries a common
    channel offset.  synth_state_dict keeps everything O(1) by construction, which never exercises what an fp16 path does with such a
    stream (the reference's autocast path normalises in fp32 first: block.py:90,93).  Done through one block's LayerScale'd MLP bias:
    x += gamma * (fc2(h) + b)  ->  b += offset / gamma on every channel, gamma[c] = 1 and b[c] = value on the outlier channels."""
    p = f"encoder.backbone.blocks.{block}."
    gam, b = sd[p + "ls2.gamma"], sd[p + "mlp.fc2.bias"]
    b += mean_offset / gam
    D = gam.numel()
    for c, v in channels:
        gam[c % D] = 1.0
        b[c % D] = v
    return sd


def save_checkpoint(path: str, cfg: dict, sd: Dict[str, torch.Tensor]) -> None:
    """Reference checkpoint format (moge/scripts/train.py:379-383)."""
    torch.save({"model_config": cfg, "model": sd}, path)


# --------------------------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------------------------
def view_plane_uv(width: int, height: int, aspect: Optional[float] = None, dtype=torch.float32) -> torch.Tensor:
    """(H, W, 2) normalised view-plane coordinates, pixel centres (geometry_torch.py:40-52)."""
    if aspect is None:
        aspect = width / height
    sx = aspect / (1 + aspect ** 2) ** 0.5
    sy = 1 / (1 + aspect ** 2) ** 0.5
    u = torch.linspace(-sx * (width - 1) / width, sx * (width - 1) / width, width, dtype=dtype)
    v = torch.linspace(-sy * (height - 1) / height, sy * (height - 1) / height, height, dtype=dtype)
    return torch.stack([u[None, :].expand(height, width), v[:, None].expand(height, width)], dim=-1)


def token_grid(height: int, width: int, num_tokens: int) -> Tuple[int, int]:
    """ViT token grid for an image (v2.py:142-147); python round() = half-to-even."""
    a = width / height
    return round((num_tokens / a) ** 0.5), round((num_tokens * a) ** 0.5)


def _intrinsics(fx: torch.Tensor, fy: torch.Tensor) -> torch.Tensor:
    """utils3d.pt.intrinsics_from_focal_center(fx, fy, 0.5, 0.5) (v2.py:266; utils3d not vendored: restated)."""
    K = torch.zeros(fx.shape + (3, 3), dtype=fx.dtype)
    K[..., 0, 0] = fx
    K[..., 1, 1] = fy
    K[..., 0, 2] = 0.5
    K[..., 1, 2] = 0.5
    K[..., 2, 2] = 1.0
    return K


def _depth_to_points(depth: torch.Tensor, K: torch.Tensor) -> torch.Tensor:
    """utils3d.pt.depth_map_to_point_map(depth, intrinsics=K) (v2.py:276): pixel-centre uv in [0,1],
    x=(u-cx)/fx*z, y=(v-cy)/fy*z (the only convention consistent with view_plane_uv + fx=focal/(2 span_x))."""
    B, H, W = depth.shape
    u = (torch.arange(W, dtype=depth.dtype) + 0.5) / W
    v = (torch.arange(H, dtype=depth.dtype) + 0.5) / H
    x = (u[None, None, :] - K[:, 0, 2, None, None]) / K[:, 0, 0, None, None] * depth
    y = (v[None, :, None] - K[:, 1, 2, None, None]) / K[:, 1, 1, None, None] * depth
    return torch.stack([x, y, depth], dim=-1)


# --------------------------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------------------------
def pos_embed_for_grid(pos_embed: torch.Tensor, h0: int, w0: int, onnx_compatible_mode: bool = False) -> torch.Tensor:
    """(1, 1+h0*w0, D) position embedding for an h0 x w0 grid (vision_transformer.py:187-221): bicubic with the
    scale_factor=(n+0.1)/37 kludge, computed in fp32; returned unchanged iff the grid is the native 37x37.
    onnx_compatible_mode (vision_transformer.py:192,202-210): always resampled, with size=(h0, w0) instead of the scale-factor kludge."""
    n = pos_embed.shape[1] - 1
    if not onnx_compatible_mode and h0 * w0 == n and h0 == w0:
        return pos_embed
    pe = pos_embed.float()
    M = int(math.sqrt(n))
    D = pe.shape[-1]
    grid = pe[:, 1:].reshape(1, M, M, D).permute(0, 3, 1, 2)
    if onnx_compatible_mode:
        grid = F.interpolate(grid, size=(h0, w0), mode="bicubic", antialias=False)
    else:
        grid = F.interpolate(grid, scale_factor=((h0 + 0.1) / M, (w0 + 0.1) / M), mode="bicubic", antialias=False)
    assert grid.shape[-2:] == (h0, w0)
    grid = grid.permute(0, 2, 3, 1).reshape(1, h0 * w0, D)
    return torch.cat([pe[:, :1], grid], dim=1).to(pos_embed.dtype)


def vit_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str, heads: int) -> torch.Tensor:
    """Pre-LN block, eval branch (block.py:110-112)."""
    B, N, D = x.shape
    h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, N, 3, heads, D // heads)
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
    a = F.scaled_dot_product_attention(q, k, v)
    a = a.permute(0, 2, 1, 3).reshape(B, N, D)
    a = F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    x = x + a * sd[p + "ls1.gamma"]
    h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h * sd[p + "ls2.gamma"]


def encoder(cfg: dict, sd: Dict[str, torch.Tensor], image: torch.Tensor, h0: int, w0: int, trace: Optional[dict] = None,
            onnx_compatible_mode: bool = False):
    """DINOv2Encoder.forward (modules.py:120-136) -> (features (B,c0,h0,w0), cls (B,D)).  onnx_compatible_mode: the resize loses its
    antialiasing (modules.py:121) and the position embedding is resampled by output size (see pos_embed_for_grid)."""
    bb = "encoder.backbone."
    D, L, heads = VIT_SPECS[cfg["encoder"]["backbone"]]
    taps = cfg["encoder"]["intermediate_layers"]
    B = image.shape[0]
    x = F.interpolate(image, (h0 * PATCH, w0 * PATCH), mode="bilinear", align_corners=False, antialias=not onnx_compatible_mode)
    x = (x - sd["encoder.image_mean"].to(x.dtype)) / sd["encoder.image_std"].to(x.dtype)
    if trace is not None:
        trace["image_14"] = x
    x = F.conv2d(x, sd[bb + "patch_embed.proj.weight"], sd[bb + "patch_embed.proj.bias"], stride=PATCH)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[bb + "cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + pos_embed_for_grid(sd[bb + "pos_embed"], h0, w0, onnx_compatible_mode)
    if trace is not None:
        trace["tokens0"] = x
    outs = []
    for i in range(L):
        x = vit_block(x, sd, f"{bb}blocks.{i}.", heads)
        if trace is not None and i == 0:
            trace["block0"] = x
        if i in taps:
            outs.append(x)
    outs = [F.layer_norm(o, (D,), sd[bb + "norm.weight"], sd[bb + "norm.bias"], 1e-6) for o in outs]
    if trace is not None:
        trace["taps"] = outs
    cls = outs[-1][:, 0]
    feats = None
    for k, o in enumerate(outs):
        f = o[:, 1:].permute(0, 2, 1).reshape(B, D, h0, w0)
        f = F.conv2d(f, sd[f"encoder.output_projections.{k}.weight"], sd[f"encoder.output_projections.{k}.bias"])
        feats = f if feats is None else feats + f
    return feats, cls


# --------------------------------------------------------------------------------------------
# decoder
# --------------------------------------------------------------------------------------------
def _conv3(x, w, b):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), w, b)


def _res_norm(x, kind, w, b):
    """in_norm / hidden_norm of ResidualConvBlock (modules.py:47-58): GroupNorm(1, C) ("layer_norm"), GroupNorm(C // 32, C) ("group_norm"),
    InstanceNorm2d(C) ("instance_norm": no affine part, instance statistics in eval mode as well - track_running_stats=False)."""
    if kind == "none":
        return x
    if kind == "instance_norm":
        return F.instance_norm(x, eps=1e-5)
    C = x.shape[1]
    return F.group_norm(x, 1 if kind == "layer_norm" else C // 32, w, b, eps=1e-5)


def _res_act(x, kind):
    """activation of ResidualConvBlock (modules.py:31-40)"""
    if kind == "relu":
        return F.relu(x)
    if kind == "leaky_relu":
        return F.leaky_relu(x, negative_slope=0.2)
    if kind == "silu":
        return F.silu(x)
    if kind == "elu":
        return F.elu(x)
    raise ValueError(f"Unsupported activation function: {kind}")


def conv_stack(sc: dict, sd: Dict[str, torch.Tensor], name: str, feats: List[Optional[torch.Tensor]]) -> List[torch.Tensor]:
    """ConvStack.forward (modules.py:242-254): residual blocks [norm ->] act -> 3x3 -> [norm ->] act -> 3x3 with every norm / activation /
    hidden width of modules.py:18-67, the x2 up-samplers conv_transpose / bilinear / nearest / pixel_shuffle (modules.py:139-181).  The released
    v2 models use no norms, ReLU, hidden = width and [conv_transpose x3, bilinear]."""
    in_norm, hid_norm = sc["res_block_in_norm"], sc["res_block_hidden_norm"]
    act = sc.get("activation", "relu")
    dims = sc["dim_res_blocks"]
    dim_out = sc["dim_out"] if isinstance(sc["dim_out"], list) else [sc["dim_out"]] * len(dims)
    outs = []
    x = None
    for l in range(len(dims)):
        f = feats[l]
        if sc["dim_in"][l] is not None:
            f = F.conv2d(f, sd[f"{name}.input_blocks.{l}.weight"], sd[f"{name}.input_blocks.{l}.bias"])
        x = f if l == 0 else x + f
        for j in range(sc["num_res_blocks"][l]):
            p = f"{name}.res_blocks.{l}.{j}.layers."
            y = _res_norm(x, in_norm, sd.get(p + "0.weight"), sd.get(p + "0.bias"))
            y = _conv3(_res_act(y, act), sd[p + "2.weight"], sd[p + "2.bias"])
            y = _res_norm(y, hid_norm, sd.get(p + "3.weight"), sd.get(p + "3.bias"))
            y = _conv3(_res_act(y, act), sd[p + "5.weight"], sd[p + "5.bias"])
            x = x + y
        if dim_out[l] is not None:
            outs.append(F.conv2d(x, sd[f"{name}.output_blocks.{l}.weight"], sd[f"{name}.output_blocks.{l}.bias"]))
        else:
            outs.append(x)
        if l < len(dims) - 1:
            p = f"{name}.resamplers.{l}."
            kind = sc["resamplers"][l]
            if kind == "pixel_shuffle":
                x = F.pixel_shuffle(_conv3(x, sd[p + "0.weight"], sd[p + "0.bias"]), 2)
                x = _conv3(x, sd[p + "2.weight"], sd[p + "2.bias"])
                continue
            if kind == "conv_transpose":
                x = F.conv_transpose2d(x, sd[p + "0.weight"], sd[p + "0.bias"], stride=2)
            elif kind == "bilinear":
                x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
            elif kind == "nearest":
                x = F.interpolate(x, scale_factor=2, mode="nearest")
            else:
                raise NotImplementedError(kind)
            x = _conv3(x, sd[p + "1.weight"], sd[p + "1.bias"])
    return outs


def remap_points(p: torch.Tensor, mode: str) -> torch.Tensor:
    """v2.py:122-136"""
    if mode == "linear":
        return p
    if mode == "sinh":
        return torch.sinh(p)
    if mode == "exp":
        z = torch.exp(p[..., 2:3])
        return torch.cat([p[..., :2] * z, z], dim=-1)
    if mode == "sinh_exp":
        return torch.cat([torch.sinh(p[..., :2]), torch.exp(p[..., 2:3])], dim=-1)
    raise ValueError(f"Invalid remap output type: {mode}")


def forward(cfg: dict, sd: Dict[str, torch.Tensor], image: torch.Tensor, num_tokens: int,
            trace: Optional[dict] = None, onnx_compatible_mode: bool = False) -> Dict[str, torch.Tensor]:
    """MoGeModel.forward (v2.py:138-192). image (B,3,H,W) in [0,1], dtype = compute dtype (fp32 / fp64)."""
    B, _, H, W = image.shape
    dt = image.dtype
    if dt != torch.float32:
        sd = {k: v.to(dt) for k, v in sd.items()}
    aspect = W / H
    h0, w0 = token_grid(H, W, num_tokens)
    feats, cls = encoder(cfg, sd, image, h0, w0, trace, onnx_compatible_mode)
    if trace is not None:
        trace["features"] = feats
        trace["cls"] = cls
    levels: List[Optional[torch.Tensor]] = [feats, None, None, None, None]
    for l in range(5):
        uv = view_plane_uv(w0 * 2 ** l, h0 * 2 ** l, aspect, dtype=dt).permute(2, 0, 1)[None].expand(B, -1, -1, -1)
        levels[l] = uv if levels[l] is None else torch.cat([levels[l], uv], dim=1)
    neck = conv_stack(cfg["neck"], sd, "neck", levels)
    if trace is not None:
        trace["neck"] = neck
    out: Dict[str, torch.Tensor] = {}
    raw = {}
    for h in HEAD_NAMES:
        if cfg.get(h) is not None:
            r = conv_stack(cfg[h], sd, h, neck)[-1]
            raw[h] = r
            out[h] = F.interpolate(r, (H, W), mode="bilinear", align_corners=False, antialias=False)
    if trace is not None:
        trace["head_raw"] = raw
    res: Dict[str, torch.Tensor] = {}
    if "points_head" in out:
        res["points"] = remap_points(out["points_head"].permute(0, 2, 3, 1), cfg.get("remap_output", "linear"))
    if "normal_head" in out:
        res["normal"] = F.normalize(out["normal_head"].permute(0, 2, 3, 1), dim=-1)
    if "mask_head" in out:
        res["mask"] = out["mask_head"].squeeze(1).sigmoid()
        if trace is not None:
            trace["mask_logit"] = out["mask_head"].squeeze(1)
    if cfg.get("scale_head") is not None:
        n = len(cfg["scale_head"]["dims"]) - 1
        s = cls
        for i in range(n):
            s = F.linear(s, sd[f"scale_head.{2 * i}.weight"], sd[f"scale_head.{2 * i}.bias"])
            if i < n - 1:
                s = F.relu(s)
        res["metric_scale"] = s.squeeze(1).exp()
    return res


# --------------------------------------------------------------------------------------------
# focal / shift recovery
# --------------------------------------------------------------------------------------------
def solve_shift(uv: np.ndarray, xyz: np.ndarray, focal: Optional[float] = None) -> Tuple[np.float32, Optional[np.float32]]:
    """geometry_numpy.py:79-112: min_shift |f*xy/(z+shift) - uv| with f closed-form (or given), LM from x0=0.

    uv (K,2) f32, xyz (K,3) f32; residual evaluated in float64 (scipy promotes through the float64 `shift`)."""
    uv64 = uv.reshape(-1, 2).astype(np.float64)
    xy = xyz[:, :2].astype(np.float64)
    z = xyz[:, 2].astype(np.float64)

    def resid(shift: float) -> np.ndarray:
        proj = xy / (z + shift)[:, None]
        f = (proj * uv64).sum() / np.square(proj).sum() if focal is None else focal
        return (f * proj - uv64).ravel()

    x, _info, _nfev = lmdif_scalar(resid, 0.0, ftol=1e-3, xtol=1e-8, gtol=1e-8, maxfev=200)
    shift = np.float32(x)
    if focal is not None:
        return shift, None
    # focal recomputed with the float32 shift in float32 numpy (geometry_numpy.py:93-94)
    proj = xyz[:, :2] / (xyz[:, 2] + shift)[:, None]
    f = (proj * uv.reshape(-1, 2)).sum() / np.square(proj).sum()
    return shift, np.float32(f)


def recover_focal_shift(points: torch.Tensor, mask: Optional[torch.Tensor], focal: Optional[torch.Tensor] = None):
    """geometry_torch.py:115-170. points (B,H,W,3) f32, mask (B,H,W) bool -> focal (B,), shift (B,) f32."""
    B, H, W, _ = points.shape
    uv = view_plane_uv(W, H, dtype=points.dtype)
    iy = torch.div(torch.arange(64) * H, 64, rounding_mode="floor")     # nearest: src = floor(dst * in / out)
    ix = torch.div(torch.arange(64) * W, 64, rounding_mode="floor")
    p_lr = points[:, iy][:, :, ix].numpy()
    uv_lr = uv[iy][:, ix].numpy()
    m_lr = None if mask is None else mask[:, iy][:, :, ix].numpy()
    f_out, s_out = [], []
    for i in range(B):
        p_i = p_lr[i].reshape(-1, 3) if m_lr is None else p_lr[i][m_lr[i]]
        uv_i = uv_lr.reshape(-1, 2) if m_lr is None else uv_lr[m_lr[i]]
        if uv_i.shape[0] < 2:
            f_out.append(1.0)
            s_out.append(0.0)
            continue
        if focal is None:
            s, f = solve_shift(uv_i, p_i)
            f_out.append(float(f))
        else:
            s, _ = solve_shift(uv_i, p_i, float(focal[i]))
        s_out.append(float(s))
    shift = torch.tensor(s_out, dtype=points.dtype)
    foc = torch.tensor(f_out, dtype=points.dtype) if focal is None else focal
    return foc, shift


@torch.inference_mode()
def infer(cfg: dict, sd: Dict[str, torch.Tensor], image: torch.Tensor, num_tokens: Optional[int] = None,
          resolution_level: int = 9, force_projection: bool = True, apply_mask: bool = True,
          fov_x=None, trace: Optional[dict] = None, onnx_compatible_mode: bool = False) -> Dict[str, torch.Tensor]:
    """MoGeModel.infer (v2.py:194-303), fp32 path (use_fp16=False)."""
    squeeze = image.dim() == 3
    if squeeze:
        image = image[None]
    image = image.float()
    H, W = image.shape[-2:]
    aspect = W / H
    if num_tokens is None:
        lo, hi = cfg["num_tokens_range"]
        num_tokens = int(lo + (resolution_level / 9) * (hi - lo))
    out = forward(cfg, sd, image, num_tokens, trace, onnx_compatible_mode)
    points, normal, mask, metric = (out.get(k) for k in ("points", "normal", "mask", "metric_scale"))
    if trace is not None:
        trace["forward"] = {k: v.clone() for k, v in out.items()}
    mask_b = mask > 0.5 if mask is not None else None
    depth = intr = None
    if points is not None:
        points = points.clone()
        if fov_x is None:
            focal, shift = recover_focal_shift(points, mask_b)
        else:
            fov = torch.as_tensor(fov_x, dtype=points.dtype)
            focal = aspect / (1 + aspect ** 2) ** 0.5 / torch.tan(torch.deg2rad(fov / 2))
            if focal.ndim == 0:
                focal = focal[None].expand(points.shape[0])
            _, shift = recover_focal_shift(points, mask_b, focal=focal)
        if trace is not None:
            trace["focal"], trace["shift"] = focal.clone(), shift.clone()
        fx = focal / 2 * (1 + aspect ** 2) ** 0.5 / aspect
        fy = focal / 2 * (1 + aspect ** 2) ** 0.5
        intr = _intrinsics(fx, fy)
        points[..., 2] += shift[:, None, None]
        if mask_b is not None:
            mask_b = mask_b & (points[..., 2] > 0)
        depth = points[..., 2].clone()
    if force_projection and depth is not None:
        points = _depth_to_points(depth, intr)
    if metric is not None:
        if points is not None:
            points = points * metric[:, None, None, None]
        if depth is not None:
            depth = depth * metric[:, None, None]
    if apply_mask and mask_b is not None:
        inf = torch.tensor(float("inf"), dtype=torch.float32)
        if points is not None:
            points = torch.where(mask_b[..., None], points, inf)
        if depth is not None:
            depth = torch.where(mask_b, depth, inf)
        if normal is not None:
            normal = torch.where(mask_b[..., None], normal, torch.zeros_like(normal))
    res = {"points": points, "intrinsics": intr, "depth": depth, "mask": mask_b, "normal": normal}
    res = {k: v for k, v in res.items() if v is not None}
    if squeeze:
        res = {k: v.squeeze(0) for k, v in res.items()}
    return res
