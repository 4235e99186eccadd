"""GPU: moge_amd.alignment (HIP kernels of csrc/alignment.hip through the C ABI) against the golden vectors of the reference's
moge/utils/alignment.py and against the numpy oracle (oracle/alignment_oracle.py) on seeded inputs.

Gates (as tests/test_alignment_oracle.py): objective value at the returned solution <= the reference's * (1 + 1e-5); solutions within 2e-3
relative (5x that for shifts: a flat optimum moves the solution more than the objective); identical ratios on the exactly-representable
fixture; size-independent properties (equivariance under scaling of the target, invariance under a permutation of the samples) at the full
evaluation size (64 x 64 samples, 3 * 4096 residuals per anchor row)."""
import os

import numpy as np
import pytest
import torch

from oracle import alignment_oracle as AO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OBJ_TOL, SOL_TOL = 1e-5, 2e-3


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moge_amd import alignment
    return alignment


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(a, b, tol, what):
    a, b = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
    assert float(err.max()) <= tol, (what, a, b)


def depth_obj(scale, shift, src, tgt, w):
    scale, shift = np.asarray(scale, np.float64), np.asarray(shift, np.float64)
    return (w * np.abs(scale[..., None] * src + shift[..., None] - tgt)).sum(-1)


def points_obj(scale, shift, src, tgt, w):
    scale, shift = np.asarray(scale, np.float64), np.asarray(shift, np.float64)
    return (w[..., None] * np.abs(scale[..., None, None] * src + shift[..., None, :] - tgt)).sum((-2, -1))


@pytest.mark.parametrize("name", ["align_l1_small", "align_l1_exact", "align_l1_long"])
def test_align_l1_golden(A, name):
    g = load(name)
    a, loss, index = A.align(dev(g["x"]), dev(g["y"]), dev(g["w"]))
    a, loss, index = a.cpu().numpy(), loss.cpu().numpy(), index.cpu().numpy()
    ref_obj = AO.objective(g["a"], g["x"], g["y"], g["w"])
    assert np.all(AO.objective(a, g["x"], g["y"], g["w"]) <= ref_obj * (1 + OBJ_TOL) + 1e-7)
    close(loss, g["loss"], 1e-4, "loss")
    rows = np.arange(g["x"].shape[0])
    sx = np.sign(g["x"])
    ratio = (g["y"] * sx) / np.maximum(g["x"] * sx, np.float32(1e-7))
    assert np.array_equal(ratio[rows, index].astype(np.float32), a), "index does not name the element whose ratio is the solution"
    if name == "align_l1_exact":
        assert np.array_equal(a, g["a"])
    else:
        close(a, g["a"], SOL_TOL, "a")


def test_align_l1_matches_oracle_bitwise_on_distinct_ratios(A):
    # float64 prefix sums on both sides and a stable order: the same element must be chosen
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1, (64, 777)).astype(np.float32)
    y = rng.normal(0, 1, (64, 777)).astype(np.float32)
    w = rng.uniform(0.1, 1, (64, 777)).astype(np.float32)
    a, loss, index = A.align(dev(x), dev(y), dev(w))
    ao, lo, io = AO.align(x, y, w)
    assert np.array_equal(index.cpu().numpy(), io)
    assert np.array_equal(a.cpu().numpy(), ao)
    close(loss, lo, 1e-5, "loss")


@pytest.mark.parametrize("name", ["align_solvers_small", "align_solvers_lr", "align_solvers_full"])
def test_solvers_golden(A, name):
    g = load(name)
    P, G, W = g["pred"], g["gt"], g["w"]
    p, q, w = dev(P), dev(G), dev(W)
    close(A.align_depth_scale(p[..., 2], q[..., 2], w), g["depth_scale"], SOL_TOL, "depth_scale")
    close(A.align_points_scale(p, q, w), g["points_scale"], SOL_TOL, "points_scale")
    close(A.align_points_z_shift(p, q, w), g["points_z_shift"], SOL_TOL, "points_z_shift")
    close(A.align_points_xyz_shift(p, q, w), g["points_xyz_shift"], SOL_TOL, "points_xyz_shift")

    s, sh = A.align_depth_affine(p[..., 2], q[..., 2], w)
    ref = depth_obj(g["depth_affine_scale"], g["depth_affine_shift"], P[..., 2], G[..., 2], W)
    assert np.all(depth_obj(s.cpu().numpy(), sh.cpu().numpy(), P[..., 2], G[..., 2], W) <= ref * (1 + OBJ_TOL))
    close(s, g["depth_affine_scale"], SOL_TOL, "depth_affine scale")
    close(sh, g["depth_affine_shift"], 5 * SOL_TOL, "depth_affine shift")

    for fn, key in ((A.align_points_scale_z_shift, "points_scale_z_shift"), (A.align_points_scale_xyz_shift, "points_scale_xyz_shift")):
        s, sh = fn(p, q, w)
        s, sh = s.cpu().numpy(), sh.cpu().numpy()
        ref = points_obj(g[key + "_scale"], g[key + "_shift"], P, G, W)
        assert np.all(points_obj(s, sh, P, G, W) <= ref * (1 + OBJ_TOL)), key
        close(s, g[key + "_scale"], SOL_TOL, key + " scale")
        assert np.abs(sh - g[key + "_shift"]).max() <= 5 * SOL_TOL * max(1.0, np.abs(g[key + "_shift"]).max()), key
        if key == "points_scale_z_shift":
            assert np.all(sh[..., :2] == 0)

    a, b = A.align_affine_lstsq(p[..., 2], 1.0 / q[..., 2])
    close(a, g["lstsq_a"], 1e-3, "lstsq a"); close(b, g["lstsq_b"], 1e-3, "lstsq b")
    a, b = A.align_affine_lstsq(p[..., 2], 1.0 / q[..., 2], w + 0.1)
    close(a, g["lstsq_w_a"], 1e-3, "lstsq_w a"); close(b, g["lstsq_w_b"], 1e-3, "lstsq_w b")


def test_full_size_properties(A):
    """64 x 64 samples, batch 2: (i) scaling the target by c scales the solution by c, (ii) permuting the samples changes nothing,
    (iii) an exact affine relation is recovered exactly up to rounding."""
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 4096
    gt = torch.rand(2, n, 3, device="cuda", generator=g) * torch.tensor([4.0, 3.0, 7.0], device="cuda") + torch.tensor([-2.0, -1.5, 0.5], device="cuda")
    pred = (gt - torch.tensor([0.1, -0.2, 0.3], device="cuda")) / 1.7 + 0.01 * torch.randn(2, n, 3, device="cuda", generator=g)
    w = 1.0 / gt.norm(dim=-1)
    w[:, ::7] = 0
    s, sh = A.align_points_scale_xyz_shift(pred, gt, w)
    s2, sh2 = A.align_points_scale_xyz_shift(pred, gt * 2.0, w)          # powers of two: exact in fp32
    assert torch.equal(s2, s * 2.0) and torch.equal(sh2, sh * 2.0)
    perm = torch.randperm(n, device="cuda", generator=g)
    s3, sh3 = A.align_points_scale_xyz_shift(pred[:, perm], gt[:, perm], w[:, perm])
    assert torch.allclose(s3, s, rtol=1e-6) and torch.allclose(sh3, sh, rtol=1e-5, atol=1e-6)
    assert torch.allclose(s, torch.full_like(s, 1.7), rtol=2e-2)
    exact = pred * 1.5 + torch.tensor([0.25, -0.5, 1.0], device="cuda")
    s4, sh4 = A.align_points_scale_xyz_shift(pred, exact, w)
    assert torch.allclose(s4, torch.full_like(s4, 1.5), rtol=1e-5) and torch.allclose(sh4, torch.tensor([0.25, -0.5, 1.0], device="cuda").expand(2, 3), atol=1e-4)


def test_edge_cases(A):
    from moge_amd._lib import MogeError
    x = torch.ones(2, 8, device="cuda")
    # all weights zero: derivative is 0 everywhere -> the first (smallest) ratio, loss 0 (alignment.py:78 searchsorted 'left')
    y = torch.arange(16, device="cuda", dtype=torch.float32).reshape(2, 8)
    a, loss, index = A.align(x, y, torch.zeros_like(x))
    assert a.tolist() == [0.0, 8.0] and loss.tolist() == [0.0, 0.0] and index.tolist() == [0, 0]
    # a single sample
    a, loss, index = A.align(torch.tensor([[2.0]], device="cuda"), torch.tensor([[-3.0]], device="cuda"), torch.ones(1, 1, device="cuda"))
    assert a.item() == -1.5 and loss.item() == 0.0 and index.item() == 0
    # negative x: signs are flipped, the ratio is what counts
    a, _, _ = A.align(torch.tensor([[-2.0, -4.0, 1.0]], device="cuda"), torch.tensor([[-6.0, -12.0, 3.0]], device="cuda"), torch.ones(1, 3, device="cuda"))
    assert a.item() == 3.0
    # broadcast batch shapes come back
    a, loss, index = A.align(torch.rand(3, 1, 50, device="cuda"), torch.rand(1, 4, 50, device="cuda"), torch.rand(50, device="cuda"))
    assert a.shape == loss.shape == index.shape == (3, 4)
    # rows longer than the LDS holds are refused, loudly
    with pytest.raises(MogeError):
        A.align(torch.rand(1, 15361, device="cuda"), torch.rand(1, 15361, device="cuda"), torch.rand(1, 15361, device="cuda"))
    a, _, _ = A.align(torch.rand(1, 15360, device="cuda"), torch.rand(1, 15360, device="cuda"), torch.rand(1, 15360, device="cuda"))
    assert torch.isfinite(a).all()
    with pytest.raises(NotImplementedError):
        A.align(x, x, x, trunc=1.0)
    with pytest.raises(RuntimeError):
        A.align(x.cpu(), x.cpu(), x.cpu())
    with pytest.raises(ValueError):
        A.align_depth_affine(x, x, torch.zeros_like(x))
