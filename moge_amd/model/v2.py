"""Host-side mirror of the reference's `moge.model.v2.MoGeModel` (moge/model/v2.py:22-303) on top of the C ABI.

Same names, argument meaning and error behaviour as the reference for the inference surface:
`from_pretrained`, `.to/.cuda/.eval/.half/.float`, `.device`, `.dtype`, `.num_tokens_range`, `forward`, `infer`.
All arithmetic happens in libmoge_hip.so; this file only resolves shapes (token grid: v2.py:142-147, default
num_tokens: v2.py:236-238), owns the output tensors and translates status codes into the reference's exceptions."""
from __future__ import annotations

import ctypes as C
import json
import os
import warnings
from numbers import Number
from pathlib import Path
from typing import IO, Any, Dict, List, Optional, Union

import torch

from .. import _lib as L

_VIT = {"dinov2_vits14": (384, 12, 6), "dinov2_vitb14": (768, 12, 12), "dinov2_vitl14": (1024, 24, 16)}
_RESAMPLERS = ["conv_transpose", "conv_transpose", "conv_transpose", "bilinear"]
_HEADS = [("points_head", L.HEAD_POINTS, 3), ("normal_head", L.HEAD_NORMAL, 3), ("mask_head", L.HEAD_MASK, 1)]


def _check_stack(name: str, sc: Dict[str, Any], dims: List[int], neck: bool):
    """-> (num_res_blocks per level, resampler codes, (in_norm, hidden_norm, activation, hidden multiplier) codes) of one ConvStack config
    (modules.py:195-240).  Supported: the x2 up-samplers conv_transpose / bilinear / nearest / pixel_shuffle per level; residual-block norms
    none / layer_norm / group_norm / instance_norm; activations relu / leaky_relu / silu / elu; dim_times_res_block_hidden >= 1.
    NOT representable: the x0.5 resamplers pixel_unshuffle / avg_pool / max_pool - MoGeModel.forward feeds level l a map of 2^l x the token
    grid (v2.py:154-160), so `x + feature` (modules.py:247-249) is a shape error in the reference itself with any of them."""
    if list(sc["dim_res_blocks"]) != list(dims):
        raise NotImplementedError(f"{name}: dim_res_blocks must equal the neck's ({dims})")
    res = sc.get("resamplers", "conv_transpose")
    res = list(res) if isinstance(res, (list, tuple)) else [res] * 4
    if len(res) != 4 or any(r not in L.RESAMPLER for r in res):
        raise NotImplementedError(f"{name}: resamplers {res} unsupported (four of {sorted(L.RESAMPLER)}; a x0.5 resampler cannot appear in a "
                                  f"working MoGe-2 decoder, v2.py:154-160)")
    in_norm, hid_norm = sc.get("res_block_in_norm", "layer_norm"), sc.get("res_block_hidden_norm", "group_norm")
    if in_norm not in L.RES_NORM or hid_norm not in L.RES_NORM:
        raise NotImplementedError(f"{name}: res_block norms ({in_norm}, {hid_norm}) unsupported ({sorted(L.RES_NORM)})")
    act = sc.get("activation", "relu")
    if act not in L.ACTIVATION:
        raise ValueError(f"Unsupported activation function: {act}")                     # modules.py:41
    mult = sc.get("dim_times_res_block_hidden", 1)
    if not isinstance(mult, int) or not 1 <= mult <= 8:
        raise NotImplementedError(f"{name}: dim_times_res_block_hidden {mult} unsupported (an integer 1 ... 8)")
    nres = sc.get("num_res_blocks", 1)
    nres = list(nres) if isinstance(nres, (list, tuple)) else [nres] * 5
    widths = (32, 64, 128, 256, 512, 1024)
    for l in range(5):
        if nres[l] > 0 and ((in_norm != "none" and dims[l] not in widths) or (hid_norm != "none" and dims[l] * mult not in widths)):
            raise NotImplementedError(f"{name}: normalised residual blocks need widths of 32 ... 1024, powers of two "
                                      f"(level {l}: {dims[l]}, hidden {dims[l] * mult})")
    dim_in = list(sc["dim_in"])
    want_in = [dims[0] + 2, 2, 2, 2, 2] if neck else list(dims)
    if dim_in != want_in:
        raise NotImplementedError(f"{name}: dim_in {dim_in} unsupported (expected {want_in})")
    dim_out = sc.get("dim_out")
    dim_out = list(dim_out) if isinstance(dim_out, (list, tuple)) else [dim_out] * 5
    if neck and any(d is not None for d in dim_out):
        raise NotImplementedError("neck.dim_out must be null")
    return nres, [L.RESAMPLER[r] for r in res], (L.RES_NORM[in_norm], L.RES_NORM[hid_norm], L.ACTIVATION[act], mult)


class MoGeModel:
    """MI355X drop-in for moge.model.v2.MoGeModel (inference only)."""

    def __init__(self, encoder: Dict[str, Any], neck: Dict[str, Any], points_head: Dict[str, Any] = None,
                 mask_head: Dict[str, Any] = None, normal_head: Dict[str, Any] = None, scale_head: Dict[str, Any] = None,
                 remap_output: str = "linear", num_tokens_range: List[int] = [1200, 3600], **deprecated_kwargs):
        if deprecated_kwargs:
            warnings.warn(f"The following deprecated/invalid arguments are ignored: {deprecated_kwargs}")
        if remap_output not in L.REMAP:
            raise ValueError(f"Invalid remap output type: {remap_output}")
        backbone = encoder["backbone"]
        if backbone not in _VIT:
            raise NotImplementedError(f"backbone {backbone} is not supported (ViT-S/B/L-14 only)")
        D, depth, heads = _VIT[backbone]
        taps = encoder["intermediate_layers"]
        taps = list(range(depth - taps, depth)) if isinstance(taps, int) else list(taps)
        dims = list(neck["dim_res_blocks"])
        if encoder["dim_out"] != dims[0] or len(dims) != 5:
            raise NotImplementedError("encoder.dim_out must equal neck.dim_res_blocks[0]; 5 levels")
        self.remap_output = remap_output
        self.num_tokens_range = list(num_tokens_range)
        self.model_config = dict(encoder=encoder, neck=neck, points_head=points_head, mask_head=mask_head,
                                 normal_head=normal_head, scale_head=scale_head, remap_output=remap_output,
                                 num_tokens_range=list(num_tokens_range))
        cfg = L.MogeConfig()
        cfg.embed_dim, cfg.depth, cfg.num_heads, cfg.n_taps = D, depth, heads, len(taps)
        for i, t in enumerate(taps):
            cfg.taps[i] = t
        neck_res, neck_rs, neck_norm = _check_stack("neck", neck, dims, True)
        head_res = head_rs = head_norm = None
        bits = 0
        self._head_names = []
        for name, bit, cout in _HEADS:
            sc = {"points_head": points_head, "normal_head": normal_head, "mask_head": mask_head}[name]
            if sc is None:
                continue
            r, rs, nm = _check_stack(name, sc, dims, False)
            do = sc.get("dim_out")
            # only the LAST level's output is used (v2.py:166 takes `[-1]`); output convs a config declares at levels 0 ... 3 produce maps the
            # reference computes and drops - their weights are accepted in the state dict and never read
            if not isinstance(do, (list, tuple)) or len(do) != 5 or do[4] != cout or any(d is not None and (not isinstance(d, int) or d <= 0) for d in do[:4]):
                raise NotImplementedError(f"{name}: dim_out must be a list of 5 ending in {cout} (levels 0 ... 3: null or a channel count)")
            if head_res is not None and (r != head_res or rs != head_rs or nm != head_norm):
                raise NotImplementedError("all heads must share num_res_blocks, resamplers and residual-block options")
            head_res, head_rs, head_norm = r, rs, nm
            bits |= bit
            self._head_names.append(name)
        if scale_head is not None:
            sd = list(scale_head["dims"])
            if len(sd) != 4 or sd[0] != D or sd[1] != sd[2] or sd[3] != 1:
                raise NotImplementedError(f"scale_head dims {sd} unsupported (expected [D, h, h, 1])")
            cfg.scale_hidden = sd[1]
            bits |= L.HEAD_SCALE
        for l in range(5):
            cfg.dims[l] = dims[l]
            cfg.neck_res_blocks[l] = neck_res[l]
            cfg.head_res_blocks[l] = (head_res or [0] * 5)[l]
        for l in range(4):
            cfg.neck_resamplers[l] = neck_rs[l]
            cfg.head_resamplers[l] = (head_rs or neck_rs)[l]
        cfg.neck_in_norm, cfg.neck_hidden_norm, cfg.neck_activation, cfg.neck_hidden_mult = neck_norm
        cfg.head_in_norm, cfg.head_hidden_norm, cfg.head_activation, cfg.head_hidden_mult = head_norm or (0, 0, 0, 1)
        cfg.heads = bits
        cfg.remap_output = L.REMAP[remap_output]
        self._cfg = cfg
        self._bits = bits
        self._state: Optional[Dict[str, torch.Tensor]] = None       # host fp32 state dict until the model is placed on a GPU
        self._blob_path: Optional[str] = None                       # or: a packed master blob on disk (from_blob), uploaded as one copy
        self._blob_offset = 0
        self._handle = None
        self._device = torch.device("cpu")
        self._dtype = torch.float32
        self._onnx_compatible_mode = False
        self.training = False
        self.sync_on_infer = True           # the reference synchronises inside recover_focal_shift; keep that contract

    # ------------------------------------------------------------------ nn.Module-like surface
    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def dtype(self) -> torch.dtype:
        return self._dtype

    @property
    def onnx_compatible_mode(self) -> bool:
        return self._onnx_compatible_mode

    @onnx_compatible_mode.setter
    def onnx_compatible_mode(self, value: bool):
        """v2.py:67-74: the forward the reference exports to ONNX (no antialiasing in the 14x resize, size-based position-embedding
        resampling).  Applied to the HIP handle now, or when the handle is created."""
        self._onnx_compatible_mode = bool(value)
        if self._handle is not None:
            L.check(L.lib.moge_set_onnx_compatible_mode(self._handle, 1 if self._onnx_compatible_mode else 0))

    def eval(self) -> "MoGeModel":
        self.training = False
        return self

    def train(self, mode: bool = True) -> "MoGeModel":
        if mode:
            raise NotImplementedError("moge_amd is inference-only")
        return self

    def requires_grad_(self, flag: bool = False) -> "MoGeModel":
        return self

    def enable_pytorch_native_sdpa(self):
        """v2.py:119-120 swaps the DINOv2 attention for torch's fused SDPA; the attention here IS a fused kernel (csrc/attention_pp.hip): nothing to do."""

    def enable_gradient_checkpointing(self):
        """v2.py:112-117: a training-memory switch with no effect on results; inference-only here, accepted and ignored."""
        warnings.warn("moge_amd is inference-only: enable_gradient_checkpointing() has no effect")

    def init_weights(self):
        """v2.py:109-110 downloads the pretrained DINOv2 backbone to START training from."""
        raise NotImplementedError("moge_amd is inference-only: load a trained checkpoint with from_pretrained()")

    def _remap_points(self, points: torch.Tensor) -> torch.Tensor:
        """v2.py:122-136 (applied inside `moge_forward` on the product path; this is the same map for callers that hold raw head outputs)."""
        if self.remap_output == "linear":
            return points
        if self.remap_output == "sinh":
            return torch.sinh(points)
        if self.remap_output == "exp":
            xy, z = points.split([2, 1], dim=-1)
            z = torch.exp(z)
            return torch.cat([xy * z, z], dim=-1)
        if self.remap_output == "sinh_exp":
            xy, z = points.split([2, 1], dim=-1)
            return torch.cat([torch.sinh(xy), torch.exp(z)], dim=-1)
        raise ValueError(f"Invalid remap output type: {self.remap_output}")

    def half(self) -> "MoGeModel":
        return self.to(torch.float16)

    def float(self) -> "MoGeModel":
        return self.to(torch.float32)

    def cuda(self, device: Optional[Union[int, torch.device]] = None) -> "MoGeModel":
        return self.to(torch.device("cuda", device if isinstance(device, int) else (device.index if device is not None else torch.cuda.current_device())))

    def to(self, *args, **kwargs) -> "MoGeModel":
        device, dtype = kwargs.get("device"), kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif a is not None:
                device = a
        if dtype is not None:
            if dtype not in (torch.float16, torch.float32):
                raise NotImplementedError(f"dtype {dtype} is not supported (float32 / float16)")
            self._dtype = dtype
        if device is not None:
            device = torch.device(device)
            if device.type == "cpu":
                raise RuntimeError("moge_amd has no CPU path: the model must live on an MI355X (device 'cuda')")
            if device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
            if self._handle is not None and device != self._device:
                self._release()
            self._device = device
            self._ensure_handle()
        if self._handle is not None and self._state_ready:
            with torch.cuda.device(self._device):
                L.check(L.lib.moge_set_precision(self._handle, L.FP16_HALF if self._dtype == torch.float16 else L.FP32, L.stream_ptr(self._device)))
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self._state or {})

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        self._state = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in state_dict.items()}
        if self._handle is not None:
            self._upload()
        return self

    def _release(self):
        if self._handle is not None:
            L.lib.moge_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    _state_ready = False

    def _ensure_handle(self):
        if self._handle is not None:
            return
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: moge_amd needs an MI355X (there is no CPU fallback)")
        h = C.c_void_p()
        L.check(self._create(h))
        self._handle = h
        self._state_ready = False
        if self._onnx_compatible_mode:
            L.check(L.lib.moge_set_onnx_compatible_mode(self._handle, 1))
        if self._state is not None:
            self._upload()
        elif self._blob_path is not None:
            self._upload_blob()

    def _create(self, h) -> int:
        return L.lib.moge_create(C.byref(self._cfg), self._device.index, C.byref(h))

    def _upload(self):
        names = [k.encode() for k in self._state]
        descs = (L.TensorDesc * len(names))()
        for i, (k, v) in enumerate(self._state.items()):
            descs[i].name = names[i]
            descs[i].data = v.data_ptr()
            descs[i].numel = v.numel()
        with torch.cuda.device(self._device):
            L.check(L.lib.moge_load_weights(self._handle, descs, len(names), L.stream_ptr(self._device)))
        self._state_ready = True

    # ------------------------------------------------------------------ packed master blob on disk
    # File = MAGIC | u64 header length | JSON header {model_config, nbytes, layout} | zero padding to 4096 | the fp32 master blob exactly
    # as the library lays it out in HBM (order = f(config) only, `build_tables` in csrc/model.hip), i.e. what the RCCL broadcast ships.
    # Loading it is one H2D copy + the on-device packing kernels: no torch.load / unpickling, no per-tensor name lookup.
    BLOB_MAGIC = b"MOGE-MI355X-MASTER-BLOB-v1\n"
    MODEL_VERSION = "v2"              # recorded in the blob header: a MoGe-1 blob (same container, moge_amd/model/v1.py) is not a MoGe-2 blob

    def save_blob(self, path: Union[str, Path]) -> None:
        """Write the device-resident fp32 master blob (+ model_config) to `path` (the model must be on a GPU with weights loaded)."""
        self._require_ready()
        blob = self.master_blob()
        header = json.dumps({"model_version": self.MODEL_VERSION, "model_config": self.model_config, "nbytes": int(blob.numel()),
                             "layout": "moge_master_blob fp32, build_tables order"}).encode()
        host = blob.cpu().numpy()
        with open(path, "wb") as f:
            f.write(self.BLOB_MAGIC)
            f.write(len(header).to_bytes(8, "little"))
            f.write(header)
            f.write(b"\0" * ((-f.tell()) % 4096))
            host.tofile(f)

    @classmethod
    def read_blob_header(cls, path: Union[str, Path]):
        with open(path, "rb") as f:
            if f.read(len(cls.BLOB_MAGIC)) != cls.BLOB_MAGIC:
                raise ValueError(f"{path}: not a moge_amd master blob")
            n = int.from_bytes(f.read(8), "little")
            header = json.loads(f.read(n).decode())
            off = f.tell()
        # blobs written before the field existed carry no version: accept them for either class (the byte size is checked against the
        # config when the blob is uploaded, _upload_blob) instead of treating a MoGe-1 blob as foreign on every load
        if header.get("model_version", cls.MODEL_VERSION) != cls.MODEL_VERSION:
            raise ValueError(f"{path}: master blob of model version {header.get('model_version')!r}, this class is {cls.MODEL_VERSION!r}")
        off += (-off) % 4096
        if os.path.getsize(path) != off + header["nbytes"]:
            raise ValueError(f"{path}: truncated master blob (expected {off + header['nbytes']} bytes)")
        return header, off

    @classmethod
    def from_blob(cls, path: Union[str, Path], model_kwargs: Optional[Dict[str, Any]] = None) -> "MoGeModel":
        """Build a model from a file written by `save_blob`; the weights go to the GPU as one copy when the model is placed there."""
        header, off = cls.read_blob_header(path)
        cfg = header["model_config"]
        if model_kwargs is not None:
            cfg.update(model_kwargs)
        try:
            model = cls(**cfg)
        except TypeError as e:              # an un-versioned (legacy) blob of the other model family: its config keywords do not fit this class
            raise ValueError(f"{path}: the blob's model_config does not describe a {cls.MODEL_VERSION} model ({e})") from e
        model._blob_path, model._blob_offset = str(path), off
        return model

    def _upload_blob(self):
        import numpy as np
        blob = self.master_blob()
        host = torch.from_numpy(np.memmap(self._blob_path, dtype=np.uint8, mode="c", offset=self._blob_offset))      # copy-on-write: file untouched
        if host.numel() != blob.numel():
            raise ValueError(f"{self._blob_path}: blob is {host.numel()} bytes, this config needs {blob.numel()} (config / library version mismatch)")
        blob.copy_(host)
        self.master_received()

    # ------------------------------------------------------------------ multi-GPU weight distribution
    def master_blob(self) -> torch.Tensor:
        """uint8 view of the fp32 master weight blob on the device (source/target of the RCCL broadcast)."""
        self._ensure_handle()
        L.check(L.lib.moge_alloc_master(self._handle))
        p, n = C.c_void_p(), C.c_size_t()
        L.check(L.lib.moge_master_blob(self._handle, C.byref(p), C.byref(n)))
        return torch.as_tensor(L.DevView(p.value, n.value), device=self._device)

    def master_received(self):
        L.check(L.lib.moge_master_ready(self._handle))
        self._state_ready = True
        self.to(self._dtype)

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: Union[str, Path, IO[bytes]], model_kwargs: Optional[Dict[str, Any]] = None,
                        **hf_kwargs) -> "MoGeModel":
        """Same contract as the reference (v2.py:76-107): local .pt path or HF repo id; checkpoint = {'model_config', 'model'}."""
        if Path(pretrained_model_name_or_path).exists():
            checkpoint_path = pretrained_model_name_or_path
        else:
            from huggingface_hub import hf_hub_download
            checkpoint_path = hf_hub_download(repo_id=pretrained_model_name_or_path, repo_type="model", filename="model.pt", **hf_kwargs)
        # sidecar cache: "<checkpoint>.mi355x-blob" written by `cache_blob=True` on a previous load, valid while the checkpoint is unchanged
        sidecar = None
        if isinstance(checkpoint_path, (str, Path)):
            sidecar = str(checkpoint_path) + ".mi355x-blob"
            try:
                if os.path.exists(sidecar) and os.path.getmtime(sidecar) >= os.path.getmtime(checkpoint_path):
                    model = cls.from_blob(sidecar, model_kwargs)
                    return model
            except (ValueError, OSError, KeyError):
                pass                                     # stale / foreign file: fall through to the checkpoint
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        model_config = checkpoint["model_config"]
        if model_kwargs is not None:
            model_config.update(model_kwargs)
        model = cls(**model_config)
        model.load_state_dict(checkpoint["model"], strict=False)
        model._sidecar = sidecar
        return model

    def cache_blob(self) -> Optional[str]:
        """Write the sidecar master blob next to the checkpoint this model was loaded from (next `from_pretrained` of the same file skips
        torch.load).  Returns the path, or None when the model did not come from a local checkpoint file."""
        sidecar = getattr(self, "_sidecar", None)
        if sidecar is None:
            return None
        self.save_blob(sidecar)
        return sidecar

    # ------------------------------------------------------------------ compute
    def _grid(self, H: int, W: int, num_tokens: int):
        aspect = W / H
        return round((num_tokens / aspect) ** 0.5), round((num_tokens * aspect) ** 0.5)        # python round: half-to-even (v2.py:147)

    def _prep_image(self, image: torch.Tensor) -> torch.Tensor:
        image = image.to(device=self._device)
        if image.dtype not in (torch.float16, torch.float32):
            image = image.float()
        # a .half() model casts the image to fp16 (v2.py:229): an fp32 image is handed over as it is with img_dtype 3 and rounded
        # to fp16 inside preprocess_kernel - bit-identical to `image.half()`, without a torch op on the hot path
        return image.contiguous()

    def _img_dtype(self, image: torch.Tensor) -> int:
        if image.dtype == torch.float16:
            return 1
        return 3 if self._dtype == torch.float16 else 0

    def _precision(self, use_fp16: bool) -> int:
        # a .half() model keeps everything in fp16, the residual stream included (scripts/infer.py:83-84; autocast is off then, v2.py:241);
        # fp32 weights + use_fp16 = torch.autocast: matrix products in fp16, LayerNorm inputs / residual adds in fp32
        if self._dtype == torch.float16:
            return L.FP16_HALF
        return L.FP16 if use_fp16 else L.FP32

    def _set_precision(self, prec: int):
        L.check(L.lib.moge_set_precision(self._handle, prec, L.stream_ptr(self._device)))

    def forward(self, image: torch.Tensor, num_tokens: int) -> Dict[str, torch.Tensor]:
        self._require_ready()
        image = self._prep_image(image)
        B, _, H, W = image.shape
        rows, cols = self._grid(H, W, int(num_tokens))
        dev = self._device
        with torch.cuda.device(dev):
            self._set_precision(L.FP16_HALF if self._dtype == torch.float16 else L.FP32)
            o = L.Outputs()
            res: Dict[str, torch.Tensor] = {}
            if self._bits & L.HEAD_POINTS:
                res["points"] = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev); o.points = res["points"].data_ptr()
            if self._bits & L.HEAD_NORMAL:
                res["normal"] = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev); o.normal = res["normal"].data_ptr()
            if self._bits & L.HEAD_MASK:
                res["mask"] = torch.empty((B, H, W), dtype=torch.float32, device=dev); o.mask_prob = res["mask"].data_ptr()
            if self._bits & L.HEAD_SCALE:
                res["metric_scale"] = torch.empty((B,), dtype=torch.float32, device=dev); o.metric_scale = res["metric_scale"].data_ptr()
            L.check(L.lib.moge_forward(self._handle, image.data_ptr(), self._img_dtype(image), B, H, W, rows, cols,
                                       C.byref(o), L.stream_ptr(dev)))
        if self._dtype == torch.float16:
            # a half model returns half tensors (v2.py:386-387): converted by the library, on the same stream
            with torch.cuda.device(dev):
                half = {k: torch.empty_like(v, dtype=torch.float16) for k, v in res.items()}
                for k, v in res.items():
                    L.check(L.lib.moge_cast_f16(v.data_ptr(), half[k].data_ptr(), v.numel(), L.stream_ptr(dev)))
            res = half
        return res

    __call__ = forward

    def _require_ready(self):
        if self._handle is None or not self._state_ready:
            raise RuntimeError("MoGeModel is not on a GPU yet (call .to('cuda')) or has no weights")

    @torch.inference_mode()
    def infer(self, image: torch.Tensor, num_tokens: int = None, resolution_level: int = 9, force_projection: bool = True,
              apply_mask: bool = True, fov_x: Optional[Union[Number, torch.Tensor]] = None, use_fp16: bool = True) -> Dict[str, torch.Tensor]:
        """Same parameters / returns as the reference `infer` (v2.py:194-303)."""
        self._require_ready()
        omit_batch_dim = image.dim() == 3
        if omit_batch_dim:
            image = image.unsqueeze(0)
        image = self._prep_image(image)
        B, _, H, W = image.shape
        return self._infer_device(image, self._img_dtype(image), B, H, W, omit_batch_dim, num_tokens, resolution_level,
                                  force_projection, apply_mask, fov_x, use_fp16)

    @torch.inference_mode()
    def infer_uint8(self, image: torch.Tensor, num_tokens: int = None, resolution_level: int = 9, force_projection: bool = True,
                    apply_mask: bool = True, fov_x: Optional[Union[Number, torch.Tensor]] = None, use_fp16: bool = True) -> Dict[str, torch.Tensor]:
        """`infer` for images as they come out of a decoder: uint8 (H, W, 3) or (B, H, W, 3), RGB.  Equivalent to the reference caller's
        `infer(torch.tensor(image / 255, dtype=torch.float32).permute(2, 0, 1), ...)` (scripts/infer.py:98-101), with the division, the
        layout change and the cast to the model dtype done on the device: a quarter of the PCIe bytes of a float upload."""
        self._require_ready()
        if image.dtype != torch.uint8 or image.shape[-1] != 3 or image.dim() not in (3, 4):
            raise ValueError("infer_uint8 expects a uint8 tensor of shape (H, W, 3) or (B, H, W, 3)")
        omit_batch_dim = image.dim() == 3
        if omit_batch_dim:
            image = image.unsqueeze(0)
        image = image.to(device=self._device, non_blocking=True).contiguous()
        B, H, W, _ = image.shape
        return self._infer_device(image, 2, B, H, W, omit_batch_dim, num_tokens, resolution_level, force_projection, apply_mask, fov_x, use_fp16)

    @torch.inference_mode()
    def depth_edge_mask(self, depth: torch.Tensor, mask: Optional[torch.Tensor] = None, rtol: float = 0.04) -> torch.Tensor:
        """`mask & ~utils3d.np.depth_map_edge(depth, rtol=rtol)` on the device (scripts/infer.py:127; default threshold = the CLI's 0.04)."""
        self._require_ready()
        squeeze = depth.dim() == 2
        d = (depth[None] if squeeze else depth).to(device=self._device, dtype=torch.float32).contiguous()
        B, H, W = d.shape
        m = None
        if mask is not None:
            m = (mask[None] if squeeze else mask).to(device=self._device, dtype=torch.bool).contiguous()
        out = torch.empty((B, H, W), dtype=torch.bool, device=self._device)
        with torch.cuda.device(self._device):
            L.check(L.lib.moge_depth_edge_mask(self._handle, d.data_ptr(), m.data_ptr() if m is not None else None, B, H, W, float(rtol),
                                               out.data_ptr(), L.stream_ptr(self._device)))
        return out[0] if squeeze else out

    def _infer_device(self, image, img_dtype, B, H, W, omit_batch_dim, num_tokens, resolution_level, force_projection, apply_mask, fov_x, use_fp16):
        if num_tokens is None:
            min_tokens, max_tokens = self.num_tokens_range
            num_tokens = int(min_tokens + (resolution_level / 9) * (max_tokens - min_tokens))
        rows, cols = self._grid(H, W, num_tokens)
        dev = self._device
        with torch.cuda.device(dev):
            self._set_precision(self._precision(use_fp16))
            o = L.Outputs()
            res: Dict[str, torch.Tensor] = {}
            if self._bits & L.HEAD_POINTS:      # every head is optional (v2.py:46-56); absent heads => absent keys (v2.py:291-298)
                res["points"] = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev); o.points = res["points"].data_ptr()
                res["intrinsics"] = torch.empty((B, 3, 3), dtype=torch.float32, device=dev); o.intrinsics = res["intrinsics"].data_ptr()
                res["depth"] = torch.empty((B, H, W), dtype=torch.float32, device=dev); o.depth = res["depth"].data_ptr()
            if self._bits & L.HEAD_MASK:
                res["mask"] = torch.empty((B, H, W), dtype=torch.bool, device=dev); o.mask = res["mask"].data_ptr()
            if self._bits & L.HEAD_NORMAL:
                res["normal"] = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev); o.normal = res["normal"].data_ptr()
            fov_ptr = None
            if fov_x is not None:
                fov = torch.as_tensor(fov_x, dtype=torch.float32, device=dev)
                if fov.ndim == 0:
                    fov = fov[None].expand(B)
                fov = fov.reshape(-1).contiguous()
                if fov.numel() != B:        # the reference indexes fov per image (IndexError there); the kernel would read past the buffer
                    raise ValueError(f"fov_x has {fov.numel()} elements for a batch of {B} images")
                fov_ptr = fov.data_ptr()
            flags = (L.FORCE_PROJECTION if force_projection else 0) | (L.APPLY_MASK if apply_mask else 0)
            L.check(L.lib.moge_infer(self._handle, image.data_ptr(), img_dtype, B, H, W, rows, cols,
                                     fov_ptr, flags, C.byref(o), L.stream_ptr(dev)))
            if self.sync_on_infer:
                L.check(L.lib.moge_sync(self._handle, L.stream_ptr(dev)))
        if omit_batch_dim:
            res = {k: v.squeeze(0) for k, v in res.items()}
        return res

    # ------------------------------------------------------------------ diagnostics
    def debug_tap(self, name: str) -> torch.Tensor:
        n = C.c_int64()
        L.check(L.lib.moge_debug_tap(self._handle, name.encode(), None, 0, C.byref(n), L.stream_ptr(self._device)))
        t = torch.empty((n.value,), dtype=torch.float32, device=self._device)
        L.check(L.lib.moge_debug_tap(self._handle, name.encode(), t.data_ptr(), n.value, C.byref(n), L.stream_ptr(self._device)))
        return t

    def profile(self, on: bool):
        L.check(L.lib.moge_profile_enable(self._handle, 1 if on else 0))

    def profile_read(self, reset: bool = True) -> Dict[str, Dict[str, float]]:
        p = L.Profile()
        L.check(L.lib.moge_profile_read(self._handle, C.byref(p), 1 if reset else 0))
        return {L.KC_NAMES[i]: dict(ms=p.ms[i], flops=p.flops[i], bytes=p.bytes[i], launches=p.launches[i]) for i in range(len(L.KC_NAMES))}
