"""CPU, world_size 2, gloo: the N>1 plumbing (sharding helpers + the broadcast/ready protocol shape) without a GPU."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from moge_amd.parallel import shard_batch, shard_mixed_shapes


def test_shard_batch_partitions():
    for n in (0, 1, 7, 32, 256):
        for w in (1, 2, 3, 8):
            got = [i for r in range(w) for i in shard_batch(n, w, r)]
            assert got == list(range(n))
            sizes = [len(shard_batch(n, w, r)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_shard_mixed_shapes_buckets_by_shape():
    shapes = [(518, 1036)] * 32 + [(1036, 518)] * 32
    seen = []
    for r in range(8):
        parts = shard_mixed_shapes(shapes, 8, r)
        assert len(parts) == 2 and all(len(p) == 4 for p in parts)
        for p in parts:
            assert len({shapes[i] for i in p}) == 1
            seen += p
    assert sorted(seen) == list(range(64))


class _FakeModel:
    """Stands in for MoGeModel on CPU: same master_blob()/master_received() protocol, host memory."""
    def __init__(self, nbytes, fill):
        self.buf = torch.full((nbytes,), fill, dtype=torch.uint8)
        self.ready = fill != 0
        self.device = torch.device("cpu")

    def master_blob(self):
        return self.buf

    def master_received(self):
        self.ready = True


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import moge_amd.parallel as P
    m = _FakeModel(1 << 16, 7 if rank == 0 else 0)
    P.broadcast_weights(m, src=0)
    ok = bool((m.buf == 7).all()) and m.ready
    # per-rank independent "inference" on its shard, then a gather of per-item results to check coverage
    items = list(P.shard_batch(10, world, rank))
    out = [None] * world
    dist.all_gather_object(out, items)
    q.put((rank, ok, out))
    dist.destroy_process_group()


def test_broadcast_protocol_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, out in res:
        assert ok, f"rank {rank} did not receive the blob"
        assert sorted(i for part in out for i in part) == list(range(10))


def test_bench_self_launch_command():
    """`python bench.py --gpus N` outside a launcher re-executes itself under torch.distributed.run (what the driver's scaling run needs):
    --print-launch shows the command without running it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--print-launch"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "4", "--warmup", "1"]


def _launch_worker_script():
    return (
        "import os, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.ones(1); dist.all_reduce(t)\n"
        "assert int(t.item()) == int(os.environ['WORLD_SIZE']) == 2\n"
        "print('RANK_OK', os.environ['RANK'], flush=True)\n"
        "dist.destroy_process_group()\n")


def test_self_launch_command_really_starts_n_ranks(tmp_path):
    """The command bench.py builds, pointed at a tiny script instead of bench.py itself, starts N ranks that can rendezvous (gloo, CPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    script = tmp_path / "w.py"
    script.write_text(_launch_worker_script())
    cmd = bench.self_launch_command([], 2)
    cmd[cmd.index(os.path.join(root, "bench.py"))] = str(script)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("RANK_OK") == 2
