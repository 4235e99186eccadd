"""bench.py - throughput of the MI355X-native `MoGeModel.infer()` hot path (driver contract).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is ONE infer() pass of the hot path over one batch of synthetic images that are already resident in HBM.
Workload (BASELINE.json metric: images/sec, moge-2-vitl 518x518 fp16): configs[2] = moge-2-vitl (no normal head),
batch 32 per GPU, torch.rand 3x518x518, default num_tokens 3600 (60x60 token grid, N=3601), fp16 weights
(model.half(): what the reference's `--fp16` is, scripts/infer.py:83-84 - fp16 residual stream, MOGE_FP16_HALF), full infer() including
focal/shift recovery and masking; outputs stay on the device.
For N>1 there is one process per GPU (torch.distributed / RCCL): rank 0 builds the synthetic checkpoint, the fp32 master
weight blob is broadcast once over xGMI, then every rank runs independent inference on its own shard (weak scaling:
per-GPU batch fixed; no steady-state collective - SURVEY.md 8(e)).  `python bench.py --gpus N` launches itself: when it is
not already running under torch.distributed.run (WORLD_SIZE unset) it re-executes as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...`.
The workload is the SAME at every N (moge-2-vitl, the config BASELINE.json's metric names, 32 images per GPU) so that the per-N values
form one weak-scaling curve; `--config moge-2-vitl-normal` is BASELINE configs[3], `--shape mixed` configs[4].

Prints ONE JSON line on rank 0.  Extra objects:
  roofline        the dominant kernel, gemm_pp128p_kernel (every launch of it and nothing else: profiler class gemm_pp): algorithmic FLOPs /
                  HIP-event time on the launch stream over the same K steps run single-stream right after the timed region; `traffic` =
                  fabric bytes per launch from the committed rocprofv3 PMC passes (moge_amd/pmc_traffic.json, default workload only; flagged
                  `traffic_stale` when csrc/gemm_pp.hip has changed since); `frac` against the 2.5 PFLOP/s datasheet peak, `frac_of_sustained`
                  against what a bare chain of the same MFMA instruction sustains on THIS box right after the run (tools/mfma_power, N=1)
  fp16_forms      the reference has two fp16 forms: `value` is model.half(); the rate of infer(use_fp16=True) on fp32 weights (= torch.autocast:
                  fp32 residual stream) is reported beside it, never as `value`
  rccl            N>1: ranks RCCL saw (all-reduce of ones), bytes and seconds of the one-time weight broadcast
  roofline_classes  the same object for the attention and decoder-conv classes (achieved / frac from the profiled pass, clock / MFMA-busy / fabric traffic
                  per launch from the committed PMC passes: moge_amd/pmc_traffic.json "classes", which names its profiles/ files)
  kernel_classes  per-class ms/step, TFLOP/s, GB/s of that profiled pass;  whole_path: end-to-end MFMA fraction
  pcie_inclusive  images/s of the caller-side pipeline (host uint8 in, all maps back to pinned host memory) - N=1 only, never `value`
  load_seconds    from_pretrained(.pt) vs from_blob(packed master blob) to a ready model
  cpu_baseline    the reference itself (kind "reference": /root/reference imported unmodified, build container only) or, where it does not
                  exist (the GPU box), the CPU oracle (kind "port": oracle/moge_oracle.py, bit-identical to the reference on the committed
                  fixtures) on single images of the same workload on this box's host cores: thread count chosen by a short calibration,
                  median of 3 after warm-up (rank 0, N=1 only)
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


def self_launch_command(argv, gpus, port=None):
    """The command `python bench.py --gpus N` turns into when it is not yet running under torch.distributed.run."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(O, cfg, sd, x1, kw, config_name, num_tokens):
    """Reference CPU path beside the GPU number: the unmodified reference when it can be imported, else the oracle.
    Thread count = the fastest of a short calibration (more threads than ~32 are SLOWER for this model in ATen); 1 warm-up (the
    calibration) + median of 3."""
    import statistics
    kind, run = "port", None
    if os.path.isdir("/root/reference/moge"):
        try:
            from oracle.make_golden import install_stubs
            install_stubs()
            from moge.model import import_model_class_by_version as ref_import
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "model.pt")
                O.save_checkpoint(path, cfg, sd)
                ref_model = ref_import("v2").from_pretrained(path).eval()
            kind, run = "reference", (lambda x, **k: ref_model.infer(x, use_fp16=False, **k))
        except Exception as e:            # noqa: BLE001 - fall back to the port, say why
            print(f"[bench] reference not importable ({e}); cpu_baseline uses the oracle", file=sys.stderr)
    if run is None:
        run = lambda x, **k: O.infer(cfg, sd, x, **k)      # noqa: E731
    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu})      # (all 256 hardware threads of the GPU box: 99 s for one image)
    calib = {}
    kcal = dict(kw); kcal["num_tokens"] = min(1369, num_tokens)
    prev = torch.get_num_threads()
    for t in cands:
        torch.set_num_threads(t)
        t1 = time.perf_counter()
        run(x1, **kcal)
        calib[t] = time.perf_counter() - t1
    best = min(calib, key=calib.get)
    torch.set_num_threads(best)
    times = []
    for _ in range(3):
        t1 = time.perf_counter()
        run(x1, **kw)
        times.append(time.perf_counter() - t1)
    torch.set_num_threads(prev)
    med = statistics.median(times)
    what = "the unmodified reference MoGeModel.infer(use_fp16=False) (stubs for the un-installed cv2 / utils3d only)" if kind == "reference" \
        else "oracle.infer (torch CPU fp32 + scalar lmdif; /root/reference does not exist on this box)"
    return {"value": round(1.0 / med, 4), "unit": "images/s", "cores": best, "kind": kind, "host_cpus": ncpu, "cpu_model": cpu_model_name(),
            "seconds_per_image": [round(t, 2) for t in times],
            "sample": f"3 single images of the same workload ({config_name}, 518x518, num_tokens {num_tokens}, fp32) through {what}; median "
                      f"{med:.1f} s; thread count {best} = fastest of a calibration at num_tokens {kcal['num_tokens']} over "
                      + ", ".join(f"{t}: {calib[t]:.1f} s" for t in cands) + " (also the warm-up)"}


def measure_sustained_mfma(seconds: float = 1.5):
    """What a bare chain of v_mfma_f32_16x16x32_f16 (random operands, no data movement) sustains on THIS box for `seconds` (tools/mfma_power):
    the chip clocks to its power limit and boxes of the pool differ by 10 % (1.78 ... 1.98 PFLOP/s seen).  -> (TFLOP/s, GHz) or None."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "mfma_power")
    try:
        if not os.path.exists(exe):
            from moge_amd.build import build_mfma_power
            build_mfma_power(verbose=False)
        env = dict(os.environ, MFMA_POWER_ONLY="1,0,512", MFMA_POWER_SECONDS=str(seconds))
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
        m = re.search(r"random data, 16x16x32, 8 waves.*?: ([0-9.]+) TF/s, shader clock ([0-9.]+) MHz", r.stdout)
        return (float(m.group(1)), float(m.group(2)) / 1e3) if m else None
    except Exception as e:      # noqa: BLE001 - a missing probe must not cost the bench line
        print(f"[bench] sustained-MFMA probe unavailable ({e})", file=sys.stderr)
        return None


class PowerSampler:
    """Socket power and shader clock of one GPU, sampled in a background thread during an UNTIMED pass of the same steps (never inside the timed
    region).  amdgpu hwmon first (reading a sysfs file costs microseconds), `rocm-smi --json` as the fall-back.  Reported as `power` in the bench
    line: avg watts, joules per image, avg shader clock - the evidence behind DESIGN.md section 7's "the chip is power-limited under this path"."""

    def __init__(self, dev_index: int, period: float = 0.05):
        import glob
        import threading
        self.period, self.samples, self._stop, self._thread = period, [], threading.Event(), None
        self.power_file = self.sclk_file = None
        self.source = self.cap_w = None
        try:
            prop = torch.cuda.get_device_properties(dev_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(prop, "pci_domain_id", 0), prop.pci_bus_id, prop.pci_device_id)
            base = "/sys/bus/pci/devices/" + bdf
            cands = glob.glob(base + "/hwmon/hwmon*/power1_average") + glob.glob(base + "/hwmon/hwmon*/power1_input")
            if cands:
                self.power_file, self.source = cands[0], "amdgpu hwmon " + cands[0].split("/")[-1] + " of " + bdf
                try:
                    self.cap_w = int(open(os.path.join(os.path.dirname(cands[0]), "power1_cap")).read().strip()) / 1e6
                except Exception:
                    pass
            if os.path.exists(base + "/pp_dpm_sclk"):
                self.sclk_file = base + "/pp_dpm_sclk"
        except Exception:
            pass
        self.dev_index = dev_index

    def _read(self):
        w = mhz = None
        if self.power_file:
            try:
                w = int(open(self.power_file).read().strip()) / 1e6
            except Exception:
                w = None
        if self.sclk_file:
            try:
                for ln in open(self.sclk_file).read().splitlines():
                    if ln.strip().endswith("*"):
                        mhz = float(ln.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
            except Exception:
                mhz = None
        if w is None:                                  # fall-back: one rocm-smi process per sample (slow: ~0.3 s each)
            import re
            import subprocess
            try:
                d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout)
                c = d[sorted(d)[min(self.dev_index, len(d) - 1)]]
                ws = [float(v) for k, v in c.items() if re.search("power", k, re.I) and re.match(r"^[0-9.]+$", str(v))]
                ms = [re.search(r"(\d+)Mhz", str(v)) for k, v in c.items() if re.search("sclk", k, re.I)]
                ms = [int(m.group(1)) for m in ms if m]
                w, mhz = (ws[0] if ws else None), (ms[0] if ms else mhz)
                self.source = "rocm-smi --showpower --showclocks"
            except Exception:
                pass
        return w, mhz

    def __enter__(self):
        import threading

        def loop():
            while not self._stop.is_set():
                self.samples.append(self._read())
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._thread.join(timeout=10)

    def summary(self, images: int, seconds: float):
        ws = [w for w, _ in self.samples if w]
        ms = [m for _, m in self.samples if m]
        if not ws:
            return None
        ws = ws[1:] if len(ws) > 3 else ws             # the first sample may predate the load
        avg = sum(ws) / len(ws)
        return {"avg_socket_w": round(avg, 1), "joules_per_image": round(avg * seconds / images, 3), "avg_sclk_mhz": round(sum(ms) / len(ms)) if ms else None,
                "power_cap_w": self.cap_w, "samples": len(ws), "seconds": round(seconds, 3), "source": self.source,
                "note": "sampled during an untimed repeat of the same steps right after the timed region (this rank's GPU)"}


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
    ap.add_argument("--no-power", action="store_true", help="skip the untimed power-sampling pass (socket watts / joules per image)")
    ap.add_argument("--no-autocast-pass", action="store_true", help="skip the extra steps in the reference's other fp16 form (fp32 weights + use_fp16); "
                                                                     "rocprofv3 passes use it so that the trace holds the headline mode's kernels only")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency leg (rocprofv3 passes that want the batch-32 launches only)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="run only the cpu_baseline leg (no GPU needed) and print it")
    ap.add_argument("--print-launch", action="store_true", help="print the torch.distributed.run command --gpus N would re-execute as, and exit")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend for N > 1 (nccl = RCCL over xGMI: production; gloo: "
                                                                               "host-staged weight broadcast, for ranks that must share a device)")
    ap.add_argument("--single-device", action="store_true", help="every rank drives cuda:0 (start-up / host-feeding test of the N-process path on a 1-GPU box; "
                                                                 "needs --backend gloo: RCCL refuses two ranks on one device).  The line says so; it is not a scaling number")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        # no GPU needed: the CPU leg alone (in the build container this times the unmodified reference: kind "reference")
        from oracle import moge_oracle as O
        cfg = O.named_configs()[args.config]
        sd = O.synth_state_dict(cfg, 0, True)
        x1 = torch.rand(1, 3, IMG, IMG, generator=torch.Generator().manual_seed(1000))
        nt = args.num_tokens or int(cfg["num_tokens_range"][1])
        print(json.dumps({"cpu_baseline": cpu_baseline(O, cfg, sd, x1, {} if args.num_tokens is None else {"num_tokens": args.num_tokens}, args.config, nt)}))
        return
    if args.print_launch:
        print(json.dumps(self_launch_command([a for a in sys.argv[1:] if a != "--print-launch"], args.gpus)))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher yet: become `python -m torch.distributed.run ... bench.py <same arguments>` (one rank per GPU over RCCL)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execvpe(sys.executable, self_launch_command(sys.argv[1:], args.gpus), env)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if args.single_device and args.backend != "gloo":
        raise SystemExit("--single-device needs --backend gloo (RCCL refuses two ranks on one device)")
    dev_index = 0 if args.single_device else local_rank
    t_start = time.perf_counter()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", dev_index)
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
    rccl = None
    if world > 1:
        torch.cuda.synchronize(dev)
        dist.barrier()
        tb = time.perf_counter()
        broadcast_weights(model, src=0)                  # one-time RCCL broadcast of the master blob
        dist.barrier()
        tb = time.perf_counter() - tb
        ones = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones)                            # how many ranks RCCL actually joined
        rccl = {"rccl_ranks": int(ones.item()), "backend": dist.get_backend(), "weight_broadcast_bytes": int(model.master_blob().numel()),
                "weight_broadcast_seconds": round(tb, 4)}
        if args.single_device:
            rccl["single_device"] = True
    torch.cuda.synchronize(dev)
    tp = time.perf_counter()
    model.half()                                         # per-rank repack of the fp32 master into the fp16 kernel layouts, on the device
    torch.cuda.synchronize(dev)
    pack_s = time.perf_counter() - tp
    if rccl is not None:
        rccl["pack_seconds_this_rank"] = round(pack_s, 4)

    B = args.batch
    default_workload = (args.batch == BATCH_PER_GPU and args.config == WORKLOAD_CONFIG and args.num_tokens is None and args.shape == f"{IMG}x{IMG}")
    g = torch.Generator(device="cpu").manual_seed(rank)            # rank 0: torch.rand(32, 3, 518, 518, manual_seed(0)) as SURVEY.md 8(d) config 3 draws it
    if args.shape == "mixed":
        xs = [torch.rand(B // 2, 3, 518, 1036, generator=g).to(dev), torch.rand(B - B // 2, 3, 1036, 518, generator=g).to(dev)]
    else:
        hh, ww = (int(v) for v in args.shape.lower().split("x"))
        xs = [torch.rand(B, 3, hh, ww, generator=g).to(dev)]
    x = xs[0]
    kw = {} if args.num_tokens is None else {"num_tokens": args.num_tokens}
    num_tokens = args.num_tokens or int(model.num_tokens_range[0] + (9 / 9) * (model.num_tokens_range[1] - model.num_tokens_range[0]))

    def barrier():
        torch.cuda.synchronize(dev)
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
    startup_s = time.perf_counter() - t_start           # process-group init + checkpoint / broadcast + packing + warm-up, this rank
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    # power pass (rank 0, untimed): the same K steps once more with the sampler thread running
    power = None
    if rank == 0 and not args.no_power:
        try:
            with PowerSampler(dev_index) as ps:
                tpw = time.perf_counter()
                for _ in range(max(args.steps, 8)):
                    step()
                torch.cuda.synchronize(dev)
                tpw = time.perf_counter() - tpw
            power = ps.summary(max(args.steps, 8) * B, tpw)
        except Exception as e:                              # noqa: BLE001  (a measurement extra must not fail the bench)
            power = {"error": str(e)[:200]}
    if world > 1:
        dist.barrier()
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
        cdev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, startup_s], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, startup_max = float(t[0].item()), float(t[1].item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, round(startup_s, 2))
        if rccl is not None:
            rccl["startup_seconds_max_over_ranks"] = round(startup_max, 2)
            rccl["startup_seconds_per_rank"] = per_rank
    assert bool(torch.isfinite(out["intrinsics"]).all())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        # B=1 latency (p50), outside the timed region
        lat = []
        x1 = x[:1].contiguous()
        for i in range(3 if args.no_latency else 12):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            model.infer(x1, **kw)
            torch.cuda.synchronize(dev)
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = sorted(lat[2:])
        # the reference's OTHER fp16 form, for transparency (never `value`): fp32 weights + use_fp16=True = torch.autocast (v2.py:241), which keeps
        # the residual stream in fp32 (MOGE_FP16) - `value` above is model.half() (scripts/infer.py:83-84), whose stream is fp16 (MOGE_FP16_HALF)
        autocast_rate = None
        if not args.no_autocast_pass:
            model.float()
            for _ in range(2):
                step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(max(3, args.steps // 2)):
                step()
            torch.cuda.synchronize(dev)
            autocast_rate = B * max(3, args.steps // 2) / (time.perf_counter() - t1)
            model.half()
        res = {
            "metric": f"images/sec (MoGeModel.infer, {args.config} {args.shape} fp16)", "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.config} infer(): batch {B}/GPU x 3x{args.shape} torch.rand, num_tokens {num_tokens}, "
                                   f"fp16 weights, synthetic checkpoint (seed 0), inputs resident in HBM, outputs left on device",
                       "global_batch": world * B, "parallelism": f"dp{world} (independent shards, one-time RCCL weight broadcast)"},
            "p50_latency_ms_batch1": round(lat[len(lat) // 2], 3),
            "fp16_forms": {"value_is": "model.half(): fp16 weights and fp16 residual stream, as the reference's `--fp16` (scripts/infer.py:83-84)",
                           "autocast_fp32_weights_images_per_s": round(autocast_rate, 3) if autocast_rate is not None else None,
                           "note": "infer(use_fp16=True) on fp32 weights = torch.autocast in the reference (v2.py:241): residual stream stays fp32; this rank only"},
        }
        if rccl is not None:
            res["rccl"] = rccl
        if power is not None:
            res["power"] = power
        if prof is not None:
            gm = prof["gemm_pp"] if prof["gemm_pp"]["launches"] else prof["gemm"]
            ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
            res["roofline"] = {"bound": "mfma", "kernel": "gemm_pp128p_kernel (persistent ping-pong GEMM): all of its launches and only those (ViT qkv / proj / fc1 / fc2 + summed out-projection, "
                                                              "v_mfma_f32_16x16x32_f16)" if prof["gemm_pp"]["launches"] else "gemm_glds_kernel / gemm_kernel (latency-regime GEMMs; no ping-pong launch in this workload)",
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
                # the PMC passes cannot run inside this process: say which kernel source they were taken at, and flag the number when
                # csrc/gemm_pp.hip has changed since (tools/pmc_traffic.py records the hash)
                import hashlib
                kpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "moge_amd", "csrc", "gemm_pp.hip")
                cur = hashlib.sha256(open(kpath, "rb").read()).hexdigest()[:16]
                res["roofline"]["traffic_taken_at"] = {"git": tj.get("git_commit"), "gemm_pp_hip_sha256_16": tj.get("gemm_pp_sha256_16")}
                if tj.get("gemm_pp_sha256_16") != cur:
                    res["roofline"]["traffic_stale"] = True
                    print(f"[bench] WARNING: moge_amd/pmc_traffic.json was measured on gemm_pp.hip {tj.get('gemm_pp_sha256_16')}, the tree has {cur}: "
                          "roofline.traffic is stale - rerun tools/profile_round.sh + tools/pmc_traffic.py", file=sys.stderr)
                if tj.get("clock_ghz"):
                    # shader clock the kernel actually held in the PMC pass (the chip clocks to its power budget: 2.4 GHz is what `peak` assumes)
                    res["roofline"]["clock_ghz"] = tj["clock_ghz"]
                    res["roofline"]["mfma_busy_at_that_clock"] = tj.get("mfma_busy_at_that_clock")
            # second ceiling: what a bare chain of the same MFMA instruction sustains on this chip under its power limit (tools/mfma_power,
            # no data movement at all) - `frac` stays against the datasheet peak
            live = measure_sustained_mfma() if world == 1 else None
            spath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "moge_amd", "mfma_sustained.json")
            if live is not None:
                res["roofline"]["sustained_peak"] = round(live[0], 1)
                res["roofline"]["frac_of_sustained"] = round(ach / live[0], 4)
                res["roofline"]["sustained_source"] = (f"tools/mfma_power on this box right after the run: bare v_mfma_f32_16x16x32_f16 chains, random operands, no data "
                                                       f"movement, held 1.5 s: {live[0]:.0f} TFLOP/s at {live[1]:.2f} GHz (boxes of the pool: 1.78 ... 1.98 PFLOP/s)")
            elif os.path.exists(spath):
                with open(spath) as f:
                    sj = json.load(f)
                res["roofline"]["sustained_peak"] = sj["tflops"]
                res["roofline"]["frac_of_sustained"] = round(ach / sj["tflops"], 4)
                res["roofline"]["sustained_source"] = sj["source"]
            # the other two MFMA classes of the step, each reproducible from this line + the named profiles/ files: achieved = algorithmic
            # FLOPs / HIP-event time of the class in the profiled pass above; clock, MFMA-busy and fabric traffic per launch from the committed rocprofv3 passes
            res["roofline_classes"] = {}
            tj_classes = {}
            if default_workload and os.path.exists(tpath):
                with open(tpath) as f:
                    tj_classes = json.load(f).get("classes", {})
            for cname, pkey, what in (("attn", "attn", "attn_pp16mq_kernel<4> (flash attention, head_dim 64; bound: MFMA + transcendental issue)"),
                                      ("conv", "conv", "conv_pp_kernel (3x3 / 4-phase resampler convs) + ConvTranspose2d-as-GEMM + 1x1 input blocks of the decoder")):
                pv = prof[pkey]
                if not pv["launches"] or pv["ms"] <= 0:
                    continue
                a = pv["flops"] / (pv["ms"] * 1e-3) / 1e12
                rc = {"bound": "mfma", "kernel": what, "achieved": round(a, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(a / PEAK_F16_TFLOPS, 4),
                      "ms_per_step": round(pv["ms"] / args.steps, 3), "launches_per_step": pv["launches"] // args.steps,
                      "algorithmic_bytes_per_launch": round(pv["bytes"] / pv["launches"]), "traffic": None}
                tc = tj_classes.get(cname)
                if tc:
                    rc.update({"traffic": round(tc["traffic_bytes"] / tc["launches"]), "traffic_unit": "bytes per launch (fabric reads x2-corrected + writes), full-size launches of the PMC pass",
                               "clock_ghz": tc.get("clock_ghz"), "mfma_busy_at_that_clock": tc.get("mfma_busy_at_that_clock"), "profiles": tc.get("files")})
                res["roofline_classes"][cname] = rc
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
            u8 = np.ascontiguousarray((x.float().cpu().permute(0, 2, 3, 1) * 255).round().clamp(0, 255).to(torch.uint8).numpy())
            pkw = {"num_tokens": num_tokens} if args.num_tokens else {}
            pipe = InferPipeline(model, B, hh, ww, use_fp16=True, **pkw)
            nb = 10
            for _ in pipe.run(iter([u8] * 3), copy=False):
                pass
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in pipe.run(iter([u8] * nb), copy=False):
                pass
            dtp = time.perf_counter() - t1
            # the resident rate AT THE SAME POINT of the run (the chip's clock drifts over tens of seconds of load: `value` above was taken
            # ~15 s earlier): same uint8 entry point, input already on the device, outputs left there
            xd8 = torch.from_numpy(u8).to(dev)
            for _ in range(2):
                model.infer_uint8(xd8, use_fp16=True, **pkw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(6):
                model.infer_uint8(xd8, use_fp16=True, **pkw)
            torch.cuda.synchronize()
            res_now = 6 * B / (time.perf_counter() - t1)
            res["pcie_inclusive"] = {"value": round(nb * B / dtp, 3), "unit": "images/s", "batches": nb,
                                     "resident_rate_measured_next_to_it": round(res_now, 3), "fraction_of_resident": round(nb * B / dtp / res_now, 4),
                                     "note": "host uint8 (B,H,W,3) -> pinned -> H2D -> infer_uint8 -> D2H of points/depth/mask/intrinsics(/normal) into "
                                             "pinned memory, 2 batches in flight (moge_amd/pipeline.py); not `value`"}
            del xd8
            del pipe
            with tempfile.TemporaryDirectory() as td:
                ckpt, blob = os.path.join(td, "model.pt"), os.path.join(td, "model.blob")
                O.save_checkpoint(ckpt, cfg, sd)
                model.save_blob(blob)
                t1 = time.perf_counter(); m_pt = MoGeModel.from_pretrained(ckpt).to(dev); torch.cuda.synchronize(); t_pt = time.perf_counter() - t1
                del m_pt
                t1 = time.perf_counter(); m_bl = MoGeModel.from_blob(blob).to(dev); torch.cuda.synchronize(); t_bl = time.perf_counter() - t1
                del m_bl
            res["load_seconds"] = {"checkpoint_pt": round(t_pt, 3), "master_blob": round(t_bl, 3), "pack_fp16_on_device": round(pack_s, 4),
                                   "note": "from_pretrained(.pt) vs from_blob(packed fp32 master blob) to a ready fp32 model on the device, page cache warm; pack_fp16_on_device = "
                                           "model.half() of the resident fp32 master (what every rank does after the one-time broadcast)"}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(O, cfg, sd, x[:1].float().cpu(), kw, args.config, num_tokens)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
