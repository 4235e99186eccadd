"""Caller-side batch driver for `MoGeModel.infer`.

The reference's caller (moge/scripts/infer.py:90-103) handles one image at a time on the default stream: decode, `/ 255` on the
host in float64, a 12 B/pixel float upload, `infer`, then five blocking `.cpu()` copies.  Here the same steps run as a three-stage
pipeline over `slots` in-flight batches:

    host uint8 (B,H,W,3) -> pinned staging -> [copy stream]  H2D, 3 B/pixel
                                              [compute stream] /255 + layout + dtype cast on the device, infer()
                                              [copy-back stream] D2H of the requested maps into pinned buffers

so PCIe transfers of batch i+1 / i-1 overlap the kernels of batch i.  torch is used for memory, streams and events only.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, Optional, Sequence

import numpy as np
import torch

from . import _lib as L


class InferPipeline:
    def __init__(self, model, batch: int, height: int, width: int, outputs: Optional[Sequence[str]] = None, slots: int = 2, **infer_kwargs):
        model._require_ready()
        self.model, self.B, self.H, self.W, self.slots = model, int(batch), int(height), int(width), int(slots)
        self.kw = infer_kwargs
        dev = model.device
        self.dev = dev
        shapes = {"points": ((self.B, self.H, self.W, 3), torch.float32), "depth": ((self.B, self.H, self.W), torch.float32),
                  "intrinsics": ((self.B, 3, 3), torch.float32)}
        if model._bits & L.HEAD_MASK:
            shapes["mask"] = ((self.B, self.H, self.W), torch.bool)
        if model._bits & L.HEAD_NORMAL:
            shapes["normal"] = ((self.B, self.H, self.W, 3), torch.float32)
        self.keys = [k for k in (outputs or shapes.keys())]
        for k in self.keys:
            if k not in shapes:
                raise KeyError(f"output {k!r} is not produced by this model (has: {sorted(shapes)})")
        self.pin_in = [torch.empty((self.B, self.H, self.W, 3), dtype=torch.uint8).pin_memory() for _ in range(self.slots)]
        self.dev_in = [torch.empty((self.B, self.H, self.W, 3), dtype=torch.uint8, device=dev) for _ in range(self.slots)]
        self.pin_out = [{k: torch.empty(shapes[k][0], dtype=shapes[k][1]).pin_memory() for k in self.keys} for _ in range(self.slots)]
        with torch.cuda.device(dev):
            self.s_h2d, self.s_comp, self.s_d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            self.e_h2d = [torch.cuda.Event() for _ in range(self.slots)]
            self.e_comp = [torch.cuda.Event() for _ in range(self.slots)]
            self.e_d2h = [torch.cuda.Event() for _ in range(self.slots)]

    def _submit(self, s: int, images: np.ndarray, n: int):
        src = torch.from_numpy(np.ascontiguousarray(images))
        self.pin_in[s][:n].copy_(src)                                   # host memcpy into the pinned slot
        with torch.cuda.stream(self.s_h2d):
            self.s_h2d.wait_event(self.e_comp[s])                       # the previous user of this device slot has been consumed
            self.dev_in[s][:n].copy_(self.pin_in[s][:n], non_blocking=True)
            self.e_h2d[s].record(self.s_h2d)
        with torch.cuda.stream(self.s_comp):
            self.s_comp.wait_event(self.e_h2d[s])
            # only the n submitted images: a short last batch must not run (or raise the sticky non-finite status for) stale slot contents
            out = self.model.infer_uint8(self.dev_in[s][:n], **self.kw)
            self.e_comp[s].record(self.s_comp)
        with torch.cuda.stream(self.s_d2h):
            self.s_d2h.wait_event(self.e_comp[s])
            for k in self.keys:
                out[k].record_stream(self.s_d2h)
                self.pin_out[s][k][:n].copy_(out[k], non_blocking=True)
            self.e_d2h[s].record(self.s_d2h)

    def _collect(self, s: int, n: int, copy: bool) -> Dict[str, np.ndarray]:
        self.e_d2h[s].synchronize()
        res = {k: self.pin_out[s][k][:n].numpy() for k in self.keys}
        return {k: v.copy() for k, v in res.items()} if copy else res

    def run(self, batches: Iterable[np.ndarray], copy: bool = True) -> Iterator[Dict[str, np.ndarray]]:
        """`batches` yields uint8 arrays (n <= B, H, W, 3); yields one dict of numpy arrays per input batch, in order.  With copy=False
        the arrays are views of the pinned slot and are valid only until the generator is advanced again (the slot is then resubmitted)."""
        sync = self.model.sync_on_infer
        self.model.sync_on_infer = False                                # no host synchronisation inside the pipeline
        pending = []                                                    # (slot, n) in submission order
        try:
            with torch.cuda.device(self.dev):
                for i, images in enumerate(batches):
                    n = int(images.shape[0])
                    if images.shape[1:] != (self.H, self.W, 3) or n > self.B or images.dtype != np.uint8:
                        raise ValueError(f"batch {i}: expected uint8 (<= {self.B}, {self.H}, {self.W}, 3), got {images.dtype} {images.shape}")
                    s = i % self.slots
                    if len(pending) == self.slots:
                        ps, pn = pending.pop(0)
                        yield self._collect(ps, pn, copy)
                    self._submit(s, images, n)
                    pending.append((s, n))
                while pending:
                    ps, pn = pending.pop(0)
                    yield self._collect(ps, pn, copy)
                L.check(L.lib.moge_sync(self.model._handle, int(self.s_comp.cuda_stream)))     # sticky device status (non-finite residuals)
        finally:
            self.model.sync_on_infer = sync
