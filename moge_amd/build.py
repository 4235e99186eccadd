"""Build libmoge_hip.so (gfx950) in-tree with hipcc.  `python -m moge_amd.build [--force]`.

The shared library travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmoge_hip.so")
SOURCES = ["gemm.hip", "gemm_pp.hip", "conv_pp.hip", "attention.hip", "attention_pp.hip", "elementwise.hip", "post.hip", "alignment.hip", "model.hip", "test_api.hip"]
EXPERIMENT_SOURCES = ["conv_rb.hip"]      # tools/experiments/: the fused residual block (slower than the two conv_pp launches; kbench / A-B only)
HEADERS = ["common.h", "launchers.h", os.path.join("..", "..", "include", "moge_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
    """experiments=True adds -DMOGE_EXPERIMENTS: the superseded / rejected kernel variants under tools/experiments/ are compiled in and become
    selectable through MOGE_PP_EXP / MOGE_ATTN_EXP / ... (tools/kbench A-B timing only; the product library is built without them)."""
    hipcc = _hipcc()
    flags = FLAGS + (["-DMOGE_EXPERIMENTS"] if experiments else [])
    stamp = os.path.join(LIBDIR, "obj", ".experiments")
    os.makedirs(os.path.join(LIBDIR, "obj"), exist_ok=True)
    if os.path.exists(stamp) != experiments:                # switching flavour rebuilds everything
        force = True
        if experiments:
            open(stamp, "w").close()
        else:
            os.remove(stamp)
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    if experiments:
        exp = os.path.join(os.path.dirname(HERE), "tools", "experiments")
        hdrs += [os.path.join(exp, f) for f in os.listdir(exp) if f.endswith(".inc")]
    jobs = []
    objs = []
    srcs = [os.path.join(CSRC, src) for src in SOURCES]
    if experiments:                                          # whole-file experiments (default-off two rounds running: out of the product library)
        srcs += [os.path.join(os.path.dirname(HERE), "tools", "experiments", f) for f in EXPERIMENT_SOURCES]
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s).replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + flags + ["-I" + CSRC, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[moge_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_tools(verbose: bool = True, experiments: bool = None) -> str:
    """tools/kbench: stand-alone kernel bench/checker linked against the in-tree library (GPU box utility).  experiments=None keeps the
    flavour the library was last built in."""
    if experiments is None:
        experiments = os.path.exists(os.path.join(LIBDIR, "obj", ".experiments"))
    lib = build(verbose=verbose, experiments=experiments)
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "kbench.hip")
    out = os.path.join(root, "tools", "kbench")
    if _stale(out, [src, lib]):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17"] + (["-DMOGE_EXPERIMENTS"] if experiments else []) + [src, "-o", out, "-L" + LIBDIR, "-lmoge_hip", "-Wl,-rpath,$ORIGIN/../moge_amd/lib"]
        if verbose:
            print("[moge_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
    return out


def build_mfma_power(verbose: bool = True) -> str:
    """tools/mfma_power: bare-MFMA sustained-rate probe (bench.py's `roofline.sustained_peak`, measured on the box the bench runs on)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "mfma_power.cpp")
    out = os.path.join(root, "tools", "mfma_power")
    if _stale(out, [src]):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", src, "-o", out]
        if verbose:
            print("[moge_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments="--experiments" in sys.argv))
    if "--tools" in sys.argv:
        print(build_tools(experiments="--experiments" in sys.argv))
