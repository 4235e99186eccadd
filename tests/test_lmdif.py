"""CPU: the scalar lmdif restatement vs scipy.optimize.least_squares(method='lm') (= MINPACK lmdif), on the
residual the reference minimises (geometry_numpy.py:79-112), well-posed and ill-posed."""
import numpy as np
import pytest
from scipy.optimize import least_squares

from oracle.lmdif import lmdif_scalar


def make_problem(seed, ill, fixed_focal):
    rng = np.random.default_rng(seed)
    K = int(rng.integers(2, 4096))
    uv = rng.uniform(-0.7, 0.7, size=(K, 2)).astype(np.float32).astype(np.float64)
    if ill:
        z = rng.normal(0.2, 1.0, size=K)            # z + shift crosses 0
        xy = rng.normal(0, 1, size=(K, 2))
    else:
        z = rng.uniform(1.0, 4.0, size=K)
        f_true, s_true = rng.uniform(0.5, 2.0), rng.uniform(-0.5, 0.5)
        xy = uv * (z + s_true)[:, None] / f_true + rng.normal(0, 0.01, size=(K, 2))
    xy = xy.astype(np.float32).astype(np.float64)
    z = z.astype(np.float32).astype(np.float64)
    focal = float(rng.uniform(0.5, 2.0)) if fixed_focal else None

    def fn(shift):
        s = float(np.asarray(shift).reshape(-1)[0])
        proj = xy / (z + s)[:, None]
        f = (proj * uv).sum() / np.square(proj).sum() if focal is None else focal
        return (f * proj - uv).ravel()
    return fn


@pytest.mark.parametrize("seed", range(24))
@pytest.mark.parametrize("ill", [False, True])
@pytest.mark.parametrize("fixed", [False, True])
def test_lmdif_matches_scipy(seed, ill, fixed):
    fn = make_problem(seed * 4 + 2 * ill + fixed, ill, fixed)
    sol = least_squares(fn, x0=0, ftol=1e-3, method="lm")
    x, info, nfev = lmdif_scalar(fn, 0.0, ftol=1e-3)
    assert nfev == sol.nfev, (nfev, sol.nfev, info, sol.status)
    # forward differences with h=1.5e-8 amplify summation-order noise on ill-conditioned problems
    tol = 1e-4
    assert abs(x - sol.x[0]) <= tol * max(abs(sol.x[0]), 1e-3), (x, sol.x[0])
