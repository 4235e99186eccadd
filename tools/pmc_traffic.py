#!/usr/bin/env python3
"""profiles/<tag>_pmc_FETCH_SIZE.csv + profiles/<tag>_pmc_WRITE_SIZE.csv -> moge_amd/pmc_traffic.json (bench.py's roofline.traffic).

Usage: python tools/pmc_traffic.py r01c          (after tools/profile_round.sh <tag> on the GPU box and copying the summaries)
Per launch of the roofline kernel class (gemm_pp128p_kernel, full-size launches only): KiB -> bytes, FETCH_SIZE doubled
(gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM section), WRITE_SIZE as reported.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    d = {}
    for ln in open(path).read().strip().splitlines()[1:]:
        p = ln.rsplit(",", 4)                       # kernel names may contain commas
        d[(p[0], int(p[1]))] = (int(p[2]), float(p[4]))
    return d


def main(tag):
    F = load(os.path.join(ROOT, "profiles", f"{tag}_pmc_FETCH_SIZE.csv"))
    W = load(os.path.join(ROOT, "profiles", f"{tag}_pmc_WRITE_SIZE.csv"))
    calls = fb = wb = 0
    for k, (c, f) in F.items():
        # full-size (batch 32) launches only: gemm_pp128p_kernel<EPK> is persistent (one workgroup per CU: 256 blocks; the batch-1 launches of
        # the latency leg have fewer tiles than CUs), gemm_pp128m16_kernel<EPK> / gemm_pp128_kernel of earlier profiles have one block per tile
        full = k[1] >= 256 if "gemm_pp128p" in k[0] else k[1] >= 1000
        if "gemm_pp128" in k[0] and full and k in W:
            calls += c
            fb += c * f * 1024 * 2
            wb += c * W[k][1] * 1024
    # effective shader clock and MFMA-busy fraction of the same launches from the SQ / GRBM pass (tools/pmc_summary.py derives both per kernel:
    # clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (that many cycles x 1024 SIMDs)), time-weighted
    clock = util = tw = 0.0
    mpath = os.path.join(ROOT, "profiles", f"{tag}_pmc_MFMA.csv")
    if os.path.exists(mpath):
        lines = open(mpath).read().strip().splitlines()
        hdr = lines[0].split(",")
        nc = len(hdr) - 1                              # columns after the kernel name
        for ln in lines[1:]:
            p = ln.rsplit(",", nc)
            rec = dict(zip(hdr[1:], p[1:]))
            if "gemm_pp128p" in p[0] and int(rec["blocks"]) >= 256 and "clock_ghz" in rec:
                w = int(rec["calls"]) * float(rec["avg_us"])
                clock += w * float(rec["clock_ghz"]); util += w * float(rec["mfma_util"]); tw += w
    # ---- the other two MFMA classes of the step: attention and the decoder convolutions, full-size (batch-32) launches only -
    # per class: launches, summed time, fabric bytes (same corrections), time-weighted clock and MFMA-busy fraction
    def cls_of(name, blocks, avg_us):
        if avg_us < 100:                                  # the batch-1 launches of the latency leg
            return None
        if "attn_pp16" in name:
            return "attn"
        if "conv_pp_kernel" in name or "gemm_pp128p_kernel<4" in name or "gemm_pp128m16_kernel<4" in name:      # <4> = EPK_CONVT: ConvTranspose2d as a GEMM
            return "conv"
        return None
    classes = {}
    mrows = {}
    if os.path.exists(mpath):
        lines = open(mpath).read().strip().splitlines()
        hdr = lines[0].split(",")
        for ln in lines[1:]:
            p = ln.rsplit(",", len(hdr) - 1)
            mrows[(p[0], int(p[1]))] = dict(zip(hdr[1:], p[1:]))
    for k, (c, f) in F.items():
        if k not in W:
            continue
        # FETCH pass: avg_us column is p[3]; reload it
        cl = None
        for ln in open(os.path.join(ROOT, "profiles", f"{tag}_pmc_FETCH_SIZE.csv")).read().strip().splitlines()[1:]:
            q = ln.rsplit(",", 4)
            if (q[0], int(q[1])) == k:
                cl = cls_of(k[0], k[1], float(q[3]))
                avg_us = float(q[3])
                break
        if not cl:
            continue
        d = classes.setdefault(cl, {"launches": 0, "us": 0.0, "fetch_bytes": 0.0, "write_bytes": 0.0, "_cw": 0.0, "_uw": 0.0, "_tw": 0.0, "kernels": []})
        d["launches"] += c; d["us"] += c * avg_us
        d["fetch_bytes"] += c * f * 1024 * 2; d["write_bytes"] += c * W[k][1] * 1024
        d["kernels"].append(f"{k[0]} x{k[1]}")
        m = mrows.get(k)
        if m and "clock_ghz" in m:
            w = int(m["calls"]) * float(m["avg_us"])
            d["_cw"] += w * float(m["clock_ghz"]); d["_uw"] += w * float(m["mfma_util"]); d["_tw"] += w
    cls_out = {}
    for cl, d in classes.items():
        cls_out[cl] = {"launches": d["launches"], "kernel_ms": round(d["us"] / 1e3, 3),
                       "traffic_bytes": round(d["fetch_bytes"] + d["write_bytes"]), "fetch_bytes": round(d["fetch_bytes"]), "write_bytes": round(d["write_bytes"]),
                       "clock_ghz": round(d["_cw"] / d["_tw"], 3) if d["_tw"] else None, "mfma_busy_at_that_clock": round(d["_uw"] / d["_tw"], 3) if d["_tw"] else None,
                       "kernels": sorted(set(d["kernels"])),
                       "files": [f"profiles/{tag}_pmc_FETCH_SIZE.csv", f"profiles/{tag}_pmc_WRITE_SIZE.csv", f"profiles/{tag}_pmc_MFMA.csv", f"profiles/{tag}_kernels_by_grid.csv"]}
    out = {
        "kernel": "gemm_pp128p_kernel", "launches": calls,
        "clock_ghz": round(clock / tw, 3) if tw else None, "mfma_busy_at_that_clock": round(util / tw, 3) if tw else None,
        "fetch_bytes_per_launch": round(fb / calls), "write_bytes_per_launch": round(wb / calls),
        "traffic_bytes_per_launch": round((fb + wb) / calls),
        "classes": cls_out,
        "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs, --kernel-trace only) of `python bench.py --steps 2 "
                  f"--warmup 1` with MOGE_BATCH_SPLIT=0; profiles/{tag}_pmc_FETCH_SIZE.csv + profiles/{tag}_pmc_WRITE_SIZE.csv; KiB -> bytes; "
                  "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are "
                  "counted, so this is fabric traffic = an upper bound on HBM bytes",
    }
    # which kernel source the passes measured (bench.py flags the number as stale when csrc/gemm_pp.hip has changed since): the hash
    # tools/profile_round.sh recorded on the GPU box, else the tree's (run this right after the profile)
    import hashlib
    import subprocess
    hpath = os.path.join(ROOT, "profiles", f"{tag}_source_hash.txt")
    if os.path.exists(hpath):
        out["gemm_pp_sha256_16"] = open(hpath).read().split()[0][:16]
    else:
        out["gemm_pp_sha256_16"] = hashlib.sha256(open(os.path.join(ROOT, "moge_amd", "csrc", "gemm_pp.hip"), "rb").read()).hexdigest()[:16]
    try:
        out["git_commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        out["git_commit"] = None
    with open(os.path.join(ROOT, "moge_amd", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01c")
