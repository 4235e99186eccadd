"""Helpers shared by the CPU (oracle) and GPU (HIP) parity tests: load a golden fixture, rebuild its inputs."""
import json
import os

import numpy as np
import torch

from oracle import moge_oracle as O
from oracle.make_golden import CASES, make_input, weights_digest

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASE_BY_NAME = {c["name"]: c for c in CASES}


def load_case(name):
    """-> (case, cfg, state_dict, input tensor, golden dict of numpy arrays, meta)"""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    case = meta["case"]
    cfg = O.named_configs()[case["config"]]
    sd = O.synth_state_dict(cfg, case["seed"], case["sane"])
    x = make_input(case)
    gold = {k: z[k] for k in z.files if k != "meta"}
    return case, cfg, sd, x, gold, meta


def subsample(name, arr, stride):
    """Apply the fixture's spatial stride to a full-resolution output (numpy)."""
    if stride <= 1 or name in ("intrinsics", "metric_scale"):
        return arr
    if name in ("points", "normal"):
        return arr[..., ::stride, ::stride, :]
    return arr[..., ::stride, ::stride]


def rel_err(a, b, floor=1.0):
    """max |a-b| / max(|b|, floor) over finite entries; the non-finite pattern must match exactly."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    fa, fb = np.isfinite(a), np.isfinite(b)
    assert (fa == fb).all(), "non-finite pattern differs"
    if not fa.any():
        return 0.0
    return float((np.abs(a[fa] - b[fa]) / np.maximum(np.abs(b[fa]), floor)).max())
