"""
Scalar (n = 1) restatement of MINPACK `lmdif` as driven by scipy.optimize.least_squares(method='lm').
TEST INFRASTRUCTURE ONLY (see oracle/moge_oracle.py header).

The reference's focal/shift recovery calls `least_squares(fn, x0=0, ftol=1e-3, method='lm')`
(moge/utils/geometry_numpy.py:90,109).  scipy (pinned 1.14.1 in the reference's requirements.txt:8) forwards that
to MINPACK `lmdif` with xtol=gtol=1e-8, maxfev=100*n*(n+1)=200, epsfcn=2.22e-16, factor=100, diag=[1.0] (mode 2)
(scipy/optimize/_lsq/least_squares.py `call_minpack`).  MINPACK is a third-party dependency that is not under
/root/reference; the algorithm below follows the published MINPACK-1 routines lmdif / fdjac2 / qrfac / lmpar /
qrsolv specialised to one unknown, and is pinned by direct comparison with scipy in tests/test_lmdif.py
(identical nfev, |dx|/|x| < 1e-6 on well- and ill-posed problems).

The HIP recovery kernel (moge_amd/csrc/recover.hip) implements exactly this state machine in fp64.
"""
from __future__ import annotations

import math
from typing import Callable, Tuple

import numpy as np

EPSMCH = 2.220446049250313e-16
DWARF = 2.2250738585072014e-308


def _enorm(v: np.ndarray) -> float:
    return float(math.sqrt(float(np.dot(v, v))))


def _lmpar1(r: float, diag: float, qtb: float, delta: float, par: float) -> Tuple[float, float]:
    """MINPACK lmpar for n=1: returns (par, x) with x the LM step solving (r^2 + par*diag^2) x = r*qtb."""
    # Gauss-Newton direction
    x = qtb / r if r != 0.0 else 0.0
    nonsing = r != 0.0
    it = 0
    dxnorm = abs(diag * x)
    fp = dxnorm - delta
    if fp <= 0.1 * delta:
        return 0.0, x
    parl = 0.0
    if nonsing:
        w = diag * (diag * x / dxnorm) / r
        temp = abs(w)
        parl = ((fp / delta) / temp) / temp
    gnorm = abs(r * qtb / diag)
    paru = gnorm / delta
    if paru == 0.0:
        paru = DWARF / min(delta, 0.1)
    par = max(par, parl)
    par = min(par, paru)
    if par == 0.0:
        par = gnorm / dxnorm
    while True:
        it += 1
        if par == 0.0:
            par = max(DWARF, 0.001 * paru)
        d = math.sqrt(par) * diag
        # qrsolv, n=1: one Givens rotation eliminating d against r
        if d == 0.0:
            sdiag = r
            wa = qtb
        else:
            if abs(r) < abs(d):
                cotan = r / d
                sin = 0.5 / math.sqrt(0.25 + 0.25 * cotan * cotan)
                cos = sin * cotan
            else:
                tan = d / r
                cos = 0.5 / math.sqrt(0.25 + 0.25 * tan * tan)
                sin = cos * tan
            sdiag = cos * r + sin * d
            wa = cos * qtb
        x = wa / sdiag if sdiag != 0.0 else 0.0
        dxnorm = abs(diag * x)
        temp = fp
        fp = dxnorm - delta
        if abs(fp) <= 0.1 * delta or (parl == 0.0 and fp <= temp and temp < 0.0) or it == 10:
            break
        w = diag * (diag * x / dxnorm) / sdiag
        t = abs(w)
        parc = ((fp / delta) / t) / t
        if fp > 0.0:
            parl = max(parl, par)
        if fp < 0.0:
            paru = min(paru, par)
        par = max(parl, par + parc)
    return par, x


def lmdif_scalar(fn: Callable[[float], np.ndarray], x0: float, ftol: float = 1e-3, xtol: float = 1e-8,
                 gtol: float = 1e-8, maxfev: int = 200, epsfcn: float = EPSMCH, factor: float = 100.0,
                 diag: float = 1.0) -> Tuple[float, int, int]:
    """Minimise |fn(x)|^2 over scalar x.  Returns (x, info, nfev) with MINPACK's info codes."""
    x = float(x0)
    fvec = np.asarray(fn(x), dtype=np.float64)
    nfev = 1
    fnorm = _enorm(fvec)
    par = 0.0
    it = 1
    info = 0
    xnorm = 0.0
    delta = 0.0
    eps = math.sqrt(max(epsfcn, EPSMCH))
    while True:
        # fdjac2: forward difference
        h = eps * abs(x)
        if h == 0.0:
            h = eps
        jac = (np.asarray(fn(x + h), dtype=np.float64) - fvec) / h
        nfev += 1
        # qrfac (one column): Householder vector v, R = rdiag
        acnorm = _enorm(jac)
        ajnorm = acnorm
        if ajnorm != 0.0:
            if jac[0] < 0.0:
                ajnorm = -ajnorm
            v = jac / ajnorm
            v[0] += 1.0
            r = -ajnorm
            # first component of Q^T fvec
            qtf = float(fvec[0] - float(np.dot(v, fvec))) if v[0] != 0.0 else float(fvec[0])
        else:
            r = 0.0
            qtf = float(fvec[0])
        if it == 1:
            xnorm = abs(diag * x)
            delta = factor * xnorm
            if delta == 0.0:
                delta = factor
        gnorm = 0.0
        if fnorm != 0.0 and acnorm != 0.0:
            gnorm = abs(r * (qtf / fnorm) / acnorm)
        if gnorm <= gtol:
            info = 4
            break
        while True:
            par, p = _lmpar1(r, diag, qtf, delta, par)
            p = -p
            x2 = x + p
            pnorm = abs(diag * p)
            if it == 1:
                delta = min(delta, pnorm)
            f2 = np.asarray(fn(x2), dtype=np.float64)
            nfev += 1
            fnorm1 = _enorm(f2)
            actred = -1.0
            if 0.1 * fnorm1 < fnorm:
                actred = 1.0 - (fnorm1 / fnorm) ** 2
            temp1 = abs(r * p) / fnorm
            temp2 = (math.sqrt(par) * pnorm) / fnorm
            prered = temp1 * temp1 + temp2 * temp2 / 0.5
            dirder = -(temp1 * temp1 + temp2 * temp2)
            ratio = actred / prered if prered != 0.0 else 0.0
            if ratio <= 0.25:
                temp = 0.5 if actred >= 0.0 else 0.5 * dirder / (dirder + 0.5 * actred)
                if 0.1 * fnorm1 >= fnorm or temp < 0.1:
                    temp = 0.1
                delta = temp * min(delta, pnorm / 0.1)
                par = par / temp
            elif par == 0.0 or ratio >= 0.75:
                delta = pnorm / 0.5
                par = 0.5 * par
            if ratio >= 1e-4:
                x = x2
                fvec = f2
                xnorm = abs(diag * x)
                fnorm = fnorm1
                it += 1
            if abs(actred) <= ftol and prered <= ftol and 0.5 * ratio <= 1.0:
                info = 1
            if delta <= xtol * xnorm:
                info = 2
            if abs(actred) <= ftol and prered <= ftol and 0.5 * ratio <= 1.0 and info == 2:
                info = 3
            if info != 0:
                break
            if nfev >= maxfev:
                info = 5
            if abs(actred) <= EPSMCH and prered <= EPSMCH and 0.5 * ratio <= 1.0:
                info = 6
            if delta <= EPSMCH * xnorm:
                info = 7
            if gnorm <= EPSMCH:
                info = 8
            if info != 0:
                break
            if ratio >= 1e-4:
                break
        if info != 0:
            break
    return x, info, nfev
