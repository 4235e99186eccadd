"""Multi-GPU plumbing for the batch-sharded inference path.

The path is embarrassingly parallel per image, so the only collective is the ONE-TIME broadcast of the fp32 master
weight blob (vitl-normal: 1.3 GB) from the rank that read the checkpoint, over RCCL/xGMI (fully connected 8-GPU mesh:
every peer has a direct link to the root).  After that every rank runs independent infer() calls on its own shard;
there is no steady-state collective and nothing to all-reduce."""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def broadcast_weights(model, src: int = 0, group=None) -> None:
    """Broadcast the master weight blob of `model` (a moge_amd MoGeModel already placed on this rank's GPU) from
    rank `src`.  Non-source ranks need no checkpoint on disk: their blob is allocated from the config alone."""
    blob = model.master_blob()                 # zero-copy uint8 view of the device buffer
    if blob.is_cuda and dist.get_backend(group) != "nccl":
        # a transport without device-memory support (gloo: the 2-ranks-on-one-GPU test, or a node without RCCL): stage through the host.
        # Production is backend "nccl" (= RCCL on ROCm): the broadcast below reads / writes the device blob directly over xGMI.
        host = blob.cpu() if dist.get_rank(group) == src else torch.empty(blob.shape, dtype=blob.dtype)
        dist.broadcast(host, src=src, group=group)
        if dist.get_rank(group) != src:
            blob.copy_(host)
    else:
        dist.broadcast(blob, src=src, group=group)
    if dist.get_rank(group) != src:
        model.master_received()
    if blob.is_cuda:
        torch.cuda.synchronize(model.device)


class RcclComm:
    """A bare RCCL communicator for hosts that do not use torch.distributed: ctypes on the librccl.so.1 of the process
    (torch's bundled copy when torch is imported, else the ROCm one).  `unique_id()` on one rank, ship the 128 bytes to the others by any
    side channel (file, socket, MPI), then `RcclComm(nranks, rank, uid)` on every rank with its GPU current."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            import ctypes as C
            for name in ("librccl.so.1", "librccl.so"):
                try:
                    cls._lib = C.CDLL(name, mode=C.RTLD_GLOBAL)
                    break
                except OSError:
                    continue
            if cls._lib is None:
                raise RuntimeError("librccl.so.1 not found")
        return cls._lib

    @classmethod
    def unique_id(cls) -> bytes:
        import ctypes as C
        buf = C.create_string_buffer(128)                     # NCCL_UNIQUE_ID_BYTES
        rc = cls.lib().ncclGetUniqueId(buf)
        if rc != 0:
            raise RuntimeError(f"ncclGetUniqueId failed ({rc})")
        return buf.raw

    def __init__(self, nranks: int, rank: int, uid: bytes):
        import ctypes as C

        class _Uid(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        u = _Uid()
        C.memmove(C.byref(u), uid, 128)
        self.ptr = C.c_void_p()
        fn = self.lib().ncclCommInitRank
        fn.argtypes = [C.POINTER(C.c_void_p), C.c_int, _Uid, C.c_int]
        rc = fn(C.byref(self.ptr), nranks, u, rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank failed ({rc})")
        self.nranks, self.rank = nranks, rank

    def destroy(self):
        if self.ptr:
            import ctypes as C
            fn = self.lib().ncclCommDestroy
            fn.argtypes = [C.c_void_p]
            fn(self.ptr)
            self.ptr = None


def broadcast_weights_rccl(model, comm: "RcclComm", root: int = 0) -> None:
    """The C-ABI form of `broadcast_weights` (moge_broadcast_weights in include/moge_hip.h): ncclBroadcast of the master blob on a bare RCCL
    communicator, no torch.distributed involved."""
    from . import _lib as L
    model._ensure_handle()
    with torch.cuda.device(model.device):
        L.check(L.lib.moge_broadcast_weights(model._handle, comm.ptr, root, L.stream_ptr(model.device)))
    if comm.rank != root:
        model._state_ready = True
        model.to(model.dtype)


def shard_batch(n_items: int, world: int, rank: int) -> range:
    """Contiguous shard [lo, hi) of a batch of n_items for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def shard_mixed_shapes(shapes: Sequence[Sequence[int]], world: int, rank: int) -> List[List[int]]:
    """BASELINE config 5 (mixed 518x1036 / 1036x518): bucket item indices by (H, W) - the ViT cannot mix shapes in one
    batch - and give every rank a contiguous shard of every bucket, so ranks stay balanced in tokens."""
    buckets = {}
    for i, s in enumerate(shapes):
        buckets.setdefault(tuple(s), []).append(i)
    out = []
    for key in sorted(buckets):
        idx = buckets[key]
        r = shard_batch(len(idx), world, rank)
        if len(r):
            out.append([idx[j] for j in r])
    return out
