"""Host-side mirror of the reference's `moge.model.v1.MoGeModel` (moge/model/v1.py:145-391) on top of the C ABI
(`moge_create_v1`, `moge_v1_forward`, `moge_v1_infer`).  Same constructor keywords as the checkpoint's `model_config`, same
`from_pretrained / forward / infer` surface and argument order as the reference; everything else (placement, precision, master blob,
profiler) is inherited from the MoGe-2 mirror - a handle is a handle."""
from __future__ import annotations

import ctypes as C
import warnings
from numbers import Number
from typing import Dict, List, Optional, Union

import torch

from .. import _lib as L
from .v2 import _VIT, MoGeModel as _MoGeModelV2


class MoGeModel(_MoGeModelV2):
    """MI355X drop-in for moge.model.v1.MoGeModel (inference only)."""

    def __init__(self, encoder: str = "dinov2_vitb14", intermediate_layers: Union[int, List[int]] = 4, dim_proj: int = 512,
                 dim_upsample: List[int] = [256, 128, 128], dim_times_res_block_hidden: int = 1, num_res_blocks: int = 1,
                 remap_output: str = "linear", res_block_norm: str = "group_norm", num_tokens_range: List[Number] = [1200, 2500],
                 last_res_blocks: int = 0, last_conv_channels: int = 32, last_conv_size: int = 1, mask_threshold: float = 0.5, **deprecated_kwargs):
        if deprecated_kwargs:
            if "trained_area_range" in deprecated_kwargs:                     # v1.py:168-171
                r = deprecated_kwargs.pop("trained_area_range")
                num_tokens_range = [r[0] // 14 ** 2, r[1] // 14 ** 2]
            warnings.warn(f"The following deprecated/invalid arguments are ignored: {deprecated_kwargs}")
        if remap_output is True:
            remap_output = "exp"
        if remap_output is False:
            remap_output = "linear"
        if remap_output not in L.REMAP:
            raise ValueError(f"Invalid remap output type: {remap_output}")
        if encoder not in _VIT:
            raise NotImplementedError(f"backbone {encoder} is not supported (ViT-S/B/L-14 only)")
        if not isinstance(last_res_blocks, int) or not 0 <= last_res_blocks <= 8 or last_conv_size not in (1, 3):
            raise NotImplementedError(f"MoGe-1 output blocks: last_res_blocks 0 ... 8 and last_conv_size 1 or 3 (got {last_res_blocks}, {last_conv_size})")
        if res_block_norm not in ("group_norm", "layer_norm"):
            raise NotImplementedError(f"res_block_norm {res_block_norm}: group_norm or layer_norm (v1.py:25)")
        if last_res_blocks > 0 and (last_conv_channels not in (32, 64) or last_conv_channels * (dim_times_res_block_hidden if isinstance(dim_times_res_block_hidden, int) else 1)
                                    not in (32, 64, 128, 256, 512, 1024)):
            raise NotImplementedError("last residual blocks need last_conv_channels 32 or 64 and a power-of-two hidden width up to 1024 (GroupNorm slab kernels)")
        if not isinstance(dim_times_res_block_hidden, int) or not 1 <= dim_times_res_block_hidden <= 8:
            raise NotImplementedError(f"dim_times_res_block_hidden {dim_times_res_block_hidden} unsupported (an integer 1 ... 8)")
        D, depth, heads = _VIT[encoder]
        taps = list(range(depth - intermediate_layers, depth)) if isinstance(intermediate_layers, int) else list(intermediate_layers)
        if not 1 <= len(dim_upsample) <= L.MOGE_V1_MAX_UP:
            raise NotImplementedError("1..4 upsample stages")
        if any(d not in (32, 64, 128, 256, 512) for d in dim_upsample):
            raise NotImplementedError(f"dim_upsample {list(dim_upsample)}: supported widths are 32, 64, 128, 256, 512 (GroupNorm slab kernels)")
        if num_res_blocks > 0 and any(d * dim_times_res_block_hidden not in (32, 64, 128, 256, 512, 1024) for d in dim_upsample):
            raise NotImplementedError(f"dim_upsample {list(dim_upsample)} x dim_times_res_block_hidden {dim_times_res_block_hidden}: "
                                      f"hidden widths must be powers of two up to 1024 (GroupNorm slab kernels)")
        self.encoder = encoder
        self.remap_output = remap_output
        self.intermediate_layers = intermediate_layers
        self.num_tokens_range = list(num_tokens_range)
        self.mask_threshold = mask_threshold
        self.model_config = dict(encoder=encoder, intermediate_layers=intermediate_layers, dim_proj=dim_proj, dim_upsample=list(dim_upsample),
                                 dim_times_res_block_hidden=dim_times_res_block_hidden, num_res_blocks=num_res_blocks, remap_output=remap_output,
                                 res_block_norm=res_block_norm, num_tokens_range=list(num_tokens_range), last_res_blocks=last_res_blocks,
                                 last_conv_channels=last_conv_channels, last_conv_size=last_conv_size, mask_threshold=mask_threshold)
        cfg = L.MogeV1Config()
        cfg.embed_dim, cfg.depth, cfg.num_heads, cfg.n_taps = D, depth, heads, len(taps)
        for i, t in enumerate(taps):
            cfg.taps[i] = t
        cfg.dim_proj, cfg.n_up = dim_proj, len(dim_upsample)
        for i, d in enumerate(dim_upsample):
            cfg.dim_upsample[i] = d
        cfg.num_res_blocks, cfg.last_conv_channels = num_res_blocks, last_conv_channels
        cfg.remap_output = L.REMAP[remap_output]
        cfg.mask_threshold = float(mask_threshold)
        cfg.last_res_blocks, cfg.last_conv_size = last_res_blocks, last_conv_size             # v1.py:103-109
        cfg.hidden_mult = dim_times_res_block_hidden                                      # v1.py:85 (configs/train/v1.json:31 trains with 2)
        cfg.res_block_norm = L.RES_NORM[res_block_norm]                                   # hidden norm: GroupNorm(Ch / 32, Ch) or GroupNorm(1, Ch), v1.py:47
        self._cfg = cfg
        self._bits = L.HEAD_POINTS | L.HEAD_MASK
        self._state = None
        self._blob_path = None
        self._blob_offset = 0
        self._handle = None
        self._device = torch.device("cpu")
        self._dtype = torch.float32
        self._onnx_compatible_mode = False
        self.training = False
        self.sync_on_infer = True

    BLOB_MAGIC = b"MOGE-MI355X-MASTER-BLOB-v1\n"
    MODEL_VERSION = "v1"              # checked by read_blob_header: a MoGe-2 blob / sidecar is rejected (ValueError -> from_pretrained falls back to the checkpoint)

    def _create(self, h) -> int:
        return L.lib.moge_create_v1(C.byref(self._cfg), self._device.index, C.byref(h))

    @property
    def onnx_compatible_mode(self) -> bool:
        return False

    @onnx_compatible_mode.setter
    def onnx_compatible_mode(self, value: bool):
        if value:
            raise NotImplementedError("MoGe-1 has no onnx_compatible_mode (docs/onnx.md covers MoGe-2 only)")

    @staticmethod
    def _resized(H: int, W: int, num_tokens: int):
        """v1.py:272-274, in Python floats exactly as the reference computes it."""
        f = ((num_tokens * 14 ** 2) / (H * W)) ** 0.5
        return int(H * f), int(W * f)

    def forward(self, image: torch.Tensor, num_tokens: int) -> Dict[str, torch.Tensor]:
        self._require_ready()
        image = self._prep_image(image)
        B, _, H, W = image.shape
        rh, rw = self._resized(H, W, int(num_tokens))
        dev = self._device
        with torch.cuda.device(dev):
            self._set_precision(L.FP16_HALF if self._dtype == torch.float16 else L.FP32)
            o = L.Outputs()
            res = {"points": torch.empty((B, H, W, 3), dtype=torch.float32, device=dev), "mask": torch.empty((B, H, W), dtype=torch.float32, device=dev)}
            o.points, o.mask_prob = res["points"].data_ptr(), res["mask"].data_ptr()
            L.check(L.lib.moge_v1_forward(self._handle, image.data_ptr(), self._img_dtype(image), B, H, W, rh, rw, C.byref(o), L.stream_ptr(dev)))
        return res

    __call__ = forward

    @torch.inference_mode()
    def infer(self, image: torch.Tensor, fov_x: Optional[Union[Number, torch.Tensor]] = None, resolution_level: int = 9, num_tokens: int = None,
              apply_mask: bool = True, force_projection: bool = True, use_fp16: bool = True) -> Dict[str, torch.Tensor]:
        """Same parameters (and order) / returns as the reference `infer` (v1.py:302-391): points, intrinsics, depth, mask."""
        self._require_ready()
        omit_batch_dim = image.dim() == 3
        if omit_batch_dim:
            image = image.unsqueeze(0)
        image = self._prep_image(image)
        B, _, H, W = image.shape
        return self._infer_v1(image, self._img_dtype(image), B, H, W, omit_batch_dim, fov_x, resolution_level, num_tokens, apply_mask, force_projection, use_fp16)

    def _infer_v1(self, image, img_dtype, B, H, W, omit_batch_dim, fov_x, resolution_level, num_tokens, apply_mask, force_projection, use_fp16):
        if num_tokens is None:
            lo, hi = self.num_tokens_range
            num_tokens = int(lo + (resolution_level / 9) * (hi - lo))
        rh, rw = self._resized(H, W, num_tokens)
        dev = self._device
        with torch.cuda.device(dev):
            self._set_precision(self._precision(use_fp16))
            o = L.Outputs()
            res = {"points": torch.empty((B, H, W, 3), dtype=torch.float32, device=dev),
                   "intrinsics": torch.empty((B, 3, 3), dtype=torch.float32, device=dev),
                   "depth": torch.empty((B, H, W), dtype=torch.float32, device=dev),
                   "mask": torch.empty((B, H, W), dtype=torch.bool, device=dev)}
            o.points, o.intrinsics, o.depth, o.mask = (res[k].data_ptr() for k in ("points", "intrinsics", "depth", "mask"))
            fov_ptr = None
            if fov_x is not None:
                fov = torch.as_tensor(fov_x, dtype=torch.float32, device=dev)
                if fov.ndim == 0:
                    fov = fov[None].expand(B)
                fov = fov.reshape(-1).contiguous()
                if fov.numel() != B:
                    raise ValueError(f"fov_x has {fov.numel()} elements for a batch of {B} images")
                fov_ptr = fov.data_ptr()
            flags = (L.FORCE_PROJECTION if force_projection else 0) | (L.APPLY_MASK if apply_mask else 0)
            L.check(L.lib.moge_v1_infer(self._handle, image.data_ptr(), img_dtype, B, H, W, rh, rw, fov_ptr, flags, C.byref(o), L.stream_ptr(dev)))
            if self.sync_on_infer:
                L.check(L.lib.moge_sync(self._handle, L.stream_ptr(dev)))
        if omit_batch_dim:
            res = {k: v.squeeze(0) for k, v in res.items()}
        return res

    @torch.inference_mode()
    def infer_uint8(self, image: torch.Tensor, fov_x: Optional[Union[Number, torch.Tensor]] = None, resolution_level: int = 9, num_tokens: int = None,
                    apply_mask: bool = True, force_projection: bool = True, use_fp16: bool = True) -> Dict[str, torch.Tensor]:
        """`infer` for uint8 (H, W, 3) / (B, H, W, 3) RGB images as a decoder produces them: what the reference's caller feeds after
        `torch.tensor(image / 255, dtype=torch.float32).permute(2, 0, 1)` (scripts/infer.py:98), with the division, layout change and model-dtype
        cast done on the device (moge_v1_infer's img_dtype 2; float32(i / 255.0) == float32(i) / float32(255) for every byte value)."""
        self._require_ready()
        if image.dtype != torch.uint8 or image.shape[-1] != 3 or image.dim() not in (3, 4):
            raise ValueError("infer_uint8 expects a uint8 tensor of shape (H, W, 3) or (B, H, W, 3)")
        omit_batch_dim = image.dim() == 3
        if omit_batch_dim:
            image = image.unsqueeze(0)
        image = image.to(device=self._device, non_blocking=True).contiguous()
        B, H, W, _ = image.shape
        return self._infer_v1(image, 2, B, H, W, omit_batch_dim, fov_x, resolution_level, num_tokens, apply_mask, force_projection, use_fp16)
