"""bench.py - throughput of the MI355X-native `MoGeModel.infer()` hot path (driver contract).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is ONE infer() pass of the hot path over one batch of synthetic images that are already resident in HBM.
Workload (BASELINE.json metric: images/sec, moge-2-vitl 518x518 fp16): configs[2] = moge-2-vitl (no normal head),
batch 32 per GPU, torch.rand 3x518x518, default num_tokens 3600 (60x60 token grid, N=3601), fp16 weights
(model.half()), full infer() including focal/shift recovery and masking; outputs stay on the device.
For N>1 the driver launches one process per GPU (torch.distributed / RCCL): rank 0 builds the synthetic checkpoint,
the fp32 master weight blob is broadcast once over xGMI, then every rank runs independent inference on its own
shard (weak scaling: per-GPU batch fixed; no steady-state collective - SURVEY.md 8(e)).

Prints ONE JSON line on rank 0.  Extra objects:
  roofline        dominant kernel class (the MFMA GEMMs of the ViT linears): algorithmic FLOPs / HIP-event time on the launch stream over the
                  same K steps run single-stream right after the timed region; `traffic` = fabric bytes per launch from the committed
                  rocprofv3 PMC passes (moge_amd/pmc_traffic.json, default workload only)
  kernel_classes  per-class ms/step, TFLOP/s, GB/s of that profiled pass;  whole_path: end-to-end MFMA fraction
  pcie_inclusive  images/s of the caller-side pipeline (host uint8 in, all maps back to pinned host memory) - N=1 only, never `value`
  load_seconds    from_pretrained(.pt) vs from_blob(packed master blob) to a ready model
  cpu_baseline    the CPU oracle (oracle/moge_oracle.py, a restatement pinned to the reference) on ONE image of the
                  same workload on this box's host cores (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F16_TFLOPS = 2500.0        # dense fp16 MFMA, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0
WORKLOAD_CONFIG = "moge-2-vitl"
BATCH_PER_GPU = 32
IMG = 518


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU)
    ap.add_argument("--config", default=WORKLOAD_CONFIG)
    ap.add_argument("--num-tokens", type=int, default=None)
    ap.add_argument("--shape", default=f"{IMG}x{IMG}", help="HxW of the synthetic images; 'mixed' = half 518x1036, half 1036x518 "
                                                           "(BASELINE configs[4]: two same-shape sub-batches per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive pipeline leg and the blob load-time comparison")
    ap.add_argument("--no-profile", action="store_true", help="disable the per-kernel HIP-event profiler")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from moge_amd.model import import_model_class_by_version
    from moge_amd.parallel import broadcast_weights
    from oracle import moge_oracle as O        # synthetic checkpoint generator + cpu_baseline leg only
    MoGeModel = import_model_class_by_version("v2")
    cfg = O.named_configs()[args.config]
    sd = None
    if rank == 0:
        sd = O.synth_state_dict(cfg, 0, True)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "model.pt")
            O.save_checkpoint(path, cfg, sd)
            model = MoGeModel.from_pretrained(path)      # the reference's loader contract
    else:
        model = MoGeModel(**cfg)
    model.to(dev).eval()
    if world > 1:
        broadcast_weights(model, src=0)                  # one-time RCCL broadcast of the master blob
    model.half()

    B = args.batch
    default_workload = (args.batch == BATCH_PER_GPU and args.config == WORKLOAD_CONFIG and args.num_tokens is None and args.shape == f"{IMG}x{IMG}")
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    if args.shape == "mixed":
        xs = [torch.rand(B // 2, 3, 518, 1036, generator=g).to(dev), torch.rand(B - B // 2, 3, 1036, 518, generator=g).to(dev)]
    else:
        hh, ww = (int(v) for v in args.shape.lower().split("x"))
        xs = [torch.rand(B, 3, hh, ww, generator=g).to(dev)]
    x = xs[0]
    kw = {} if args.num_tokens is None else {"num_tokens": args.num_tokens}
    num_tokens = args.num_tokens or int(model.num_tokens_range[0] + (9 / 9) * (model.num_tokens_range[1] - model.num_tokens_range[0]))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def step():
        o = None
        for xi in xs:               # one step = one infer() per same-shape sub-batch (the reference cannot mix shapes in a tensor)
            o = model.infer(xi, **kw)
        return o

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel-class HIP-event profile: the SAME K steps again with events around every launch (on the launch stream).
    # The profiler serialises the two half-batch streams of the production path (a kernel's event bracket must see only that
    # kernel), so it runs right after the timed region instead of inside it; its own step time is reported alongside.
    prof = None
    prof_ms_per_step = None
    if not args.no_profile:
        model.profile(True)
        model.profile_read(reset=True)
        step()
        model.profile_read(reset=True)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        prof_ms_per_step = (time.perf_counter() - t1) / args.steps * 1e3
        prof = model.profile_read(reset=True)
        model.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(out["intrinsics"]).all())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        # B=1 latency (p50), outside the timed region
        lat = []
        x1 = x[:1].contiguous()
        for i in range(12):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            model.infer(x1, **kw)
            torch.cuda.synchronize(dev)
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = sorted(lat[2:])
        res = {
            "metric": f"images/sec (MoGeModel.infer, {args.config} {args.shape} fp16)", "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.config} infer(): batch {B}/GPU x 3x{args.shape} torch.rand, num_tokens {num_tokens}, "
                                   f"fp16 weights, synthetic checkpoint (seed 0), inputs resident in HBM, outputs left on device",
                       "global_batch": world * B, "parallelism": f"dp{world} (independent shards, one-time RCCL weight broadcast)"},
            "p50_latency_ms_batch1": round(lat[len(lat) // 2], 3),
        }
        if prof is not None:
            gm = prof["gemm"]
            ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
            res["roofline"] = {"bound": "mfma", "kernel": "gemm_pp128m16_kernel (ViT qkv/proj/fc1/fc2 + out-proj ping-pong MFMA GEMMs, v_mfma_f32_16x16x32_f16)",
                               "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_TFLOPS, 4),
                               "traffic": None, "avg_launch_ms": round(gm["ms"] / max(gm["launches"], 1), 4), "launches": gm["launches"],
                               "algorithmic_bytes_per_launch": round(gm["bytes"] / max(gm["launches"], 1))}
            # HBM-side bytes per launch come from separate rocprofv3 --pmc passes (they cannot be collected inside this process);
            # the committed summary applies to the default workload only and names its source
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "moge_amd", "pmc_traffic.json")
            if default_workload and os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                res["roofline"]["traffic"] = tj["traffic_bytes_per_launch"]
                res["roofline"]["traffic_unit"] = "bytes per launch (fabric reads x2-corrected + writes)"
                res["roofline"]["traffic_source"] = tj["source"]
            tot_ms = sum(v["ms"] for v in prof.values())
            tot_fl = sum(v["flops"] for v in prof.values())
            res["kernel_classes"] = {k: {"ms_per_step": round(v["ms"] / args.steps, 3),
                                         "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 and v["flops"] > 0 else None,
                                         "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 and v["bytes"] > 0 else None}
                                     for k, v in prof.items()}
            res["whole_path"] = {"algorithmic_tflop_per_image": round(tot_fl / args.steps / B / 1e12, 4),
                                 "mfma_frac_of_peak_end_to_end": round(tot_fl / args.steps / (ms_per_step * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                                 "kernel_ms_per_step": round(tot_ms / args.steps, 3),
                                 "profiled_pass_ms_per_step": round(prof_ms_per_step, 3),
                                 "note": "kernel classes / roofline: HIP events on the launch stream over K single-stream steps run "
                                         "right after the timed region (same inputs); the timed region runs the production path "
                                         "(two half-batch streams, no events)"}
        if world == 1 and not args.no_pcie and args.shape != "mixed":
            # PCIe-inclusive rate of the caller-side pipeline (never `value`): uint8 host batches in, all maps back to pinned host memory,
            # transfers of neighbouring batches overlapped with the kernels (moge_amd/pipeline.py); also the load-time comparison of the
            # packed master blob against the .pt checkpoint
            import numpy as np
            from moge_amd.pipeline import InferPipeline
            hh, ww = x.shape[-2:]
            u8 = (x.float().cpu().permute(0, 2, 3, 1) * 255).round().clamp(0, 255).to(torch.uint8).numpy()
            pipe = InferPipeline(model, B, hh, ww, use_fp16=True, **({"num_tokens": num_tokens} if args.num_tokens else {}))
            nb = 6
            for _ in pipe.run(iter([u8] * 2), copy=False):
                pass
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in pipe.run(iter([u8] * nb), copy=False):
                pass
            dtp = time.perf_counter() - t1
            res["pcie_inclusive"] = {"value": round(nb * B / dtp, 3), "unit": "images/s", "batches": nb,
                                     "note": "host uint8 (B,H,W,3) -> pinned -> H2D -> infer_uint8 -> D2H of points/depth/mask/intrinsics(/normal) into "
                                             "pinned memory, 2 batches in flight (moge_amd/pipeline.py); not `value`"}
            del pipe
            with tempfile.TemporaryDirectory() as td:
                ckpt, blob = os.path.join(td, "model.pt"), os.path.join(td, "model.blob")
                O.save_checkpoint(ckpt, cfg, sd)
                model.save_blob(blob)
                t1 = time.perf_counter(); m_pt = MoGeModel.from_pretrained(ckpt).to(dev); torch.cuda.synchronize(); t_pt = time.perf_counter() - t1
                del m_pt
                t1 = time.perf_counter(); m_bl = MoGeModel.from_blob(blob).to(dev); torch.cuda.synchronize(); t_bl = time.perf_counter() - t1
                del m_bl
            res["load_seconds"] = {"checkpoint_pt": round(t_pt, 3), "master_blob": round(t_bl, 3),
                                   "note": "from_pretrained(.pt) vs from_blob(packed fp32 master blob) to a ready fp32 model on the device, page cache warm"}
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample: ONE image of the same workload through the CPU oracle (fp32), all host threads
            xc = x[:1].float().cpu()
            t1 = time.perf_counter()
            O.infer(cfg, sd, xc, **kw)
            dt = time.perf_counter() - t1
            res["cpu_baseline"] = {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"1 image of the same workload ({args.config}, 518x518, num_tokens {num_tokens}) through oracle.infer "
                                             f"(torch CPU fp32 + scalar lmdif), {dt:.1f} s"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
