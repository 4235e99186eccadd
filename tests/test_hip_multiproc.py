"""GPU: the N>1 path with REAL models.  A gpurun box has one GPU, and RCCL refuses two ranks on one device, so the two ranks share cuda:0 and
the process group is gloo: `broadcast_weights` then stages the blob through the host, everything else is the production path - rank 1 builds
its model from the config alone (no checkpoint), receives rank 0's fp32 master blob into `master_blob()`, calls `master_received()`, and its
infer() output on its own shard must equal what rank 0 computes for the same images, bit for bit.  (The RCCL transport itself is exercised by
`python bench.py --gpus N`, which reports `rccl.rccl_ranks`.)"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ckpt, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moge_amd.model import import_model_class_by_version
    from moge_amd.parallel import broadcast_weights, shard_batch
    from oracle import moge_oracle as O
    M = import_model_class_by_version("v2")
    cfg = O.named_configs()["tiny-vits-normal"]
    model = (M.from_pretrained(ckpt) if rank == 0 else M(**cfg)).to("cuda:0").eval()
    broadcast_weights(model, src=0)
    x = torch.rand(6, 3, 84, 112, generator=torch.Generator().manual_seed(9))
    mine = list(shard_batch(6, world, rank))
    model.half()
    out = model.infer(x[mine], num_tokens=108)
    # numpy arrays travel through the queue BY VALUE; torch tensors would be handed over as file descriptors that the parent can only
    # fetch while this process is still alive (a worker that finishes first made the test fail with EOFError in rebuild_storage_fd)
    res = {k: v.cpu().numpy() for k, v in out.items()}
    if rank == 0:
        full = {k: v.cpu().numpy() for k, v in model.infer(x, num_tokens=108).items()}      # what one process computes for the whole batch
        q.put(("full", full))
    q.put((rank, mine, res))
    dist.barrier()
    dist.destroy_process_group()


def test_real_model_blob_broadcast_and_sharded_infer_world2(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    from oracle import moge_oracle as O
    cfg = O.named_configs()["tiny-vits-normal"]
    ckpt = str(tmp_path / "model.pt")
    O.save_checkpoint(ckpt, cfg, O.synth_state_dict(cfg, 0, True))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ckpt, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(3)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = {k: torch.from_numpy(v) for k, v in next(g[1] for g in got if g[0] == "full").items()}
    shards = [(g[0], g[1], {k: torch.from_numpy(v) for k, v in g[2].items()}) for g in got if g[0] != "full"]
    assert sorted(i for _, mine, _ in shards for i in mine) == list(range(6))
    for rank, mine, res in shards:
        for k, v in res.items():
            ref = full[k][mine]
            if v.dtype == torch.bool:
                assert torch.equal(v, ref), (rank, k)
            else:
                fin = torch.isfinite(ref)
                assert torch.equal(fin, torch.isfinite(v)) and torch.equal(v[fin], ref[fin]), f"rank {rank}: {k} differs from the single-process result"
