"""GPU: the N>1 path with REAL models.  A gpurun box has one GPU, and RCCL refuses two ranks on one device, so the two ranks share cuda:0 and
the process group is gloo: `broadcast_weights` then stages the blob through the host, everything else is the production path - rank 1 builds
its model from the config alone (no checkpoint), receives rank 0's fp32 master blob into `master_blob()`, calls `master_received()`, and its
infer() output on its own shard must equal what rank 0 computes for the same images, bit for bit.  (The RCCL transport itself is exercised by
`python bench.py --gpus N`, which reports `rccl.rccl_ranks`.)"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, rdzv, ckpt, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + rdzv, rank=rank, world_size=world)      # file rendezvous: no port to pick, none to collide on
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
    procs = [ctx.Process(target=_worker, args=(r, 2, str(tmp_path / "rdzv"), ckpt, q)) for r in range(2)]
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


# ---- the RCCL transport itself, on the one GPU a test box has (the nccl branch of broadcast_weights had never run) -------------
def _nccl_world1(rdzv, ckpt, q):
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="file://" + rdzv, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from moge_amd.model import import_model_class_by_version
    from moge_amd.parallel import RcclComm, broadcast_weights, broadcast_weights_rccl
    M = import_model_class_by_version("v2")
    model = M.from_pretrained(ckpt).to("cuda:0").eval()
    x = torch.rand(2, 3, 84, 112, generator=torch.Generator().manual_seed(9))
    before = {k: v.cpu().numpy() for k, v in model.infer(x, num_tokens=108).items()}
    blob0 = model.master_blob().clone()
    # (1) torch.distributed, backend nccl (= RCCL): the zero-copy DevView tensor of the master blob goes through ncclBroadcast
    broadcast_weights(model, src=0)
    ones = torch.ones(1, device="cuda:0")
    dist.all_reduce(ones)
    ok_blob = bool(torch.equal(model.master_blob(), blob0))
    # (2) the C ABI (moge_broadcast_weights) on a bare communicator of the same RCCL instance
    comm = RcclComm(1, 0, RcclComm.unique_id())
    broadcast_weights_rccl(model, comm, root=0)
    comm.destroy()
    ok_blob = ok_blob and bool(torch.equal(model.master_blob(), blob0))
    after = {k: v.cpu().numpy() for k, v in model.infer(x, num_tokens=108).items()}
    q.put(dict(backend=dist.get_backend(), ranks=int(ones.item()), blob_unchanged=ok_blob, nbytes=int(blob0.numel()),
               same=all((before[k] == after[k])[~(before[k] != before[k])].all() if before[k].dtype.kind == "f" else (before[k] == after[k]).all() for k in before)))
    dist.destroy_process_group()


def test_rccl_broadcast_paths_run_on_the_gpu_world1(tmp_path):
    """backend "nccl" IS RCCL on ROCm.  One rank (a test box has one GPU and RCCL refuses two ranks on a device): the broadcast is a self-copy,
    but every piece of the production path executes - RCCL initialises on the device, `broadcast_weights` takes its nccl branch with the
    zero-copy view of the master blob, `moge_broadcast_weights` (C ABI) resolves ncclBroadcast from the RCCL instance in the process and runs it
    on a bare communicator - and the model computes the same outputs afterwards."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    from oracle import moge_oracle as O
    cfg = O.named_configs()["tiny-vits-normal"]
    ckpt = str(tmp_path / "model.pt")
    O.save_checkpoint(ckpt, cfg, O.synth_state_dict(cfg, 0, True))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1, args=(str(tmp_path / "rdzv"), ckpt, q))
    p.start()
    got = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    print("[rccl world-1]", got)
    assert got["backend"] == "nccl" and got["ranks"] == 1 and got["blob_unchanged"] and got["same"], got


def test_eight_worker_processes_start_and_feed_one_gpu(tmp_path):
    """`bench.py --gpus 8` as the driver launches it (torch.distributed.run, one rank per process), with all eight ranks on the ONE GPU of
    the test box (--single-device, gloo transport: RCCL refuses ranks that share a device) and a tiny model: the N-process path - rendezvous,
    rank 0's checkpoint load, the blob broadcast to seven config-only ranks, per-rank packing, warm-up, the barrier-bracketed timed region,
    the max-over-ranks reduction, rank 0's JSON line - runs end to end, and start-up does not serialise across ranks."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    import re
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--config", "tiny-vits-normal", "--batch", "4", "--shape", "84x112", "--num-tokens", "108", "--steps", "3", "--warmup", "1",
              "--no-cpu-baseline", "--no-pcie", "--no-profile"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(n):
        # `--standalone`: the launcher's own store binds port 0 and hands the number to the ranks - no "pick a free port, close it, pass it on"
        # window in which someone else can take it; every rank's stdout AND stderr is kept (--tee 3)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
               "--tee", "3", os.path.join(root, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--single-device"] + common
        t = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
        wall = time.perf_counter() - t
        if r.returncode != 0:                             # the ranks' own output, not only the launcher's summary: a failure here must be READ, not retried
            lines = [ln for ln in (r.stdout + "\n" + r.stderr).splitlines() if "amdgpu.ids" not in ln and "hostname of the client socket" not in ln]
            raise AssertionError("bench.py --gpus %d failed (exit %d):\n%s" % (n, r.returncode, "\n".join(lines[-200:])))
        out = [re.sub(r"^\[\w+\]:", "", ln) for ln in r.stdout.splitlines()]      # (--tee prefixes every line with its rank tag)
        line = [ln for ln in out if ln.startswith("{")][-1]
        return json.loads(line), wall

    # No retry: round 4 saw ONE unexplained loss of rank 1 in ~15 launches, with the rank's message hidden behind the launcher's
    # summary.  The launch now keeps every rank's output and takes its port from the launcher; if it fails again the log says why.
    d, w8 = run(8)
    print("[8 processes, 1 GPU]", json.dumps({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "rccl")}), f"wall {w8:.1f} s")
    assert d["n_gpus"] == 8 and d["rccl"]["rccl_ranks"] == 8 and d["rccl"]["backend"] == "gloo" and d["rccl"]["single_device"]
    assert d["config"]["global_batch"] == 32 and d["value"] > 0
    # eight ranks start concurrently: the slowest rank's start-up (import + rendezvous + load / broadcast + packing + warm-up) stays far below
    # eight sequential start-ups (one is ~10-20 s, dominated by `import torch` on a cold box)
    assert d["rccl"]["startup_seconds_max_over_ranks"] < 120, d["rccl"]
    per_rank = d["rccl"]["startup_seconds_per_rank"]
    print("[8 processes, 1 GPU] start-up seconds per rank:", per_rank)
    assert len(per_rank) == 8 and max(per_rank) < 2.5 * min(per_rank) + 10, per_rank          # no rank waits for another's start-up
