"""CPU: oracle/alignment_oracle.py (numpy restatement) against the golden vectors of the reference's moge/utils/alignment.py
(tests/golden/align_*.npz, written by oracle/make_golden_alignment.py from the reference itself).

What is compared (see the oracle's header): the OBJECTIVE value at the returned solution against the reference's, relative 1e-5 (two correct
implementations may pick different members of a near-tie, the objective agrees to rounding); the solution itself with a looser tolerance;
indices exactly on the exactly-representable fixture."""
import os

import numpy as np
import pytest

from oracle import alignment_oracle as AO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OBJ_TOL = 1e-5          # relative, objective value
SOL_TOL = 2e-3          # relative, scale / shift (a near-flat optimum moves the solution more than the objective)


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def depth_obj(scale, shift, src, tgt, w):
    return (w * np.abs(scale[..., None] * src + shift[..., None] - tgt)).sum(-1)


def points_obj(scale, shift, src, tgt, w):
    return (w[..., None] * np.abs(scale[..., None, None] * src + shift[..., None, :] - tgt)).sum((-2, -1))


def close(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
    assert float(err.max()) <= tol, (what, a, b)


@pytest.mark.parametrize("name", ["align_l1_small", "align_l1_exact", "align_l1_long"])
def test_align_l1(name):
    g = load(name)
    a, loss, index = AO.align(g["x"], g["y"], g["w"])
    ref_obj = AO.objective(g["a"], g["x"], g["y"], g["w"])
    my_obj = AO.objective(a, g["x"], g["y"], g["w"])
    assert np.all(my_obj <= ref_obj * (1 + OBJ_TOL) + 1e-7), (my_obj, ref_obj)
    close(loss, g["loss"], 1e-4, "loss")
    if name == "align_l1_exact":
        assert np.array_equal(a, g["a"])
        # the chosen element may be any member of a tie in y/x; its ratio must be the reference's
        x, y = g["x"], g["y"]
        rows = np.arange(x.shape[0])
        assert np.array_equal((y / x)[rows, index], (y / x)[rows, g["index"]])
    else:
        close(a, g["a"], SOL_TOL, "a")


@pytest.mark.parametrize("name", ["align_solvers_small", "align_solvers_lr", "align_solvers_full"])
def test_solvers(name):
    g = load(name)
    P, G, W = g["pred"], g["gt"], g["w"]
    close(AO.align_depth_scale(P[..., 2], G[..., 2], W), g["depth_scale"], SOL_TOL, "depth_scale")
    close(AO.align_points_scale(P, G, W), g["points_scale"], SOL_TOL, "points_scale")
    close(AO.align_points_z_shift(P, G, W), g["points_z_shift"], SOL_TOL, "points_z_shift")
    close(AO.align_points_xyz_shift(P, G, W), g["points_xyz_shift"], SOL_TOL, "points_xyz_shift")

    s, sh = AO.align_depth_affine(P[..., 2], G[..., 2], W)
    ref = depth_obj(g["depth_affine_scale"], g["depth_affine_shift"], P[..., 2], G[..., 2], W)
    assert np.all(depth_obj(s, sh, P[..., 2], G[..., 2], W) <= ref * (1 + OBJ_TOL))
    close(s, g["depth_affine_scale"], SOL_TOL, "depth_affine scale")
    close(sh, g["depth_affine_shift"], 5 * SOL_TOL, "depth_affine shift")

    for fn, key in ((AO.align_points_scale_z_shift, "points_scale_z_shift"), (AO.align_points_scale_xyz_shift, "points_scale_xyz_shift")):
        s, sh = fn(P, G, W)
        ref = points_obj(g[key + "_scale"], g[key + "_shift"], P, G, W)
        assert np.all(points_obj(s, sh, P, G, W) <= ref * (1 + OBJ_TOL)), key
        close(s, g[key + "_scale"], SOL_TOL, key + " scale")
        assert np.abs(sh - g[key + "_shift"]).max() <= 5 * SOL_TOL * max(1.0, np.abs(g[key + "_shift"]).max()), key

    a, b = AO.align_affine_lstsq(P[..., 2], 1.0 / G[..., 2])
    close(a, g["lstsq_a"], 1e-3, "lstsq a"); close(b, g["lstsq_b"], 1e-3, "lstsq b")
    a, b = AO.align_affine_lstsq(P[..., 2], 1.0 / G[..., 2], W + 0.1)
    close(a, g["lstsq_w_a"], 1e-3, "lstsq_w a"); close(b, g["lstsq_w_b"], 1e-3, "lstsq_w b")


def test_scatter_min_last_on_ties():
    mn, idx = AO.scatter_min(3, np.array([0, 0, 2, 0]), np.array([2.0, 1.0, 5.0, 1.0], np.float32))
    assert mn[0] == 1.0 and idx[0] == 3 and np.isinf(mn[1]) and idx[1] == -1 and idx[2] == 2
