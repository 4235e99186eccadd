#!/usr/bin/env python3
"""Phase timeline of the two-workgroups-per-CU GEMM (tools/experiments gemm_pp_kernel<4,2,2,*,3>, PP_EXP 6; PP_EXP 5 = its one-workgroup 256x256 sibling): every workgroup's
start / main loop begin / main loop end / end stamps (s_memtime ticks - the unit tools/kbench's KB_TS lines print as clocks) + HW_ID / XCC_ID from `KB_TS=1 tools/kbench gemm <shape>` (CSV).  Workgroups are grouped by CU;
for every workgroup the time of its main loop is split by what its CU-mates were doing: nobody else resident, a mate in ITS main loop, a mate in prologue / epilogue.
A least-squares fit of  K-phases = r_alone T_alone + r_both T_both + r_pe T_pe  over all workgroups gives the main-loop rate in each situation - i.e. what a main loop gains
or loses while the other workgroup's epilogue runs beside it.   usage: python tools/pp64_timeline.py <csv> <K>"""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
K = int(sys.argv[2]); nph = K // 32
T = 1.0   # one tick; durations below are printed in k ticks
wgs = []
for r in rows:
    t0, t1, t2, t3 = (int(r[k]) for k in ("t_start", "t_main_begin", "t_main_end", "t_end"))
    if not t3: continue
    hw = int(r["hw_id"]); xcc = int(r["xcc_id"]) & 0xf
    cu = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)      # (XCC, SE, SH, CU)
    wgs.append((cu, t0, t1, t2, t3))
by = {}
for w in wgs: by.setdefault(w[0], []).append(w)
print(f"{len(wgs)} workgroups on {len(by)} CUs; durations in k ticks of s_memtime")
d = np.array([[w[2] - w[1], w[3] - w[2], w[4] - w[3]] for w in wgs], dtype=float) * T / 1e3
print("per workgroup (k ticks): prologue %.2f  main loop %.2f  epilogue + store drain %.2f  (medians; total %.2f)" % (*np.median(d, axis=0), np.median(d.sum(axis=1))))
A, y = [], []
occ = {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
span = 0.0
for cu, lst in by.items():
    ev = sorted(set(t for w in lst for t in w[1:]))
    for a, b in zip(ev[:-1], ev[1:]):
        mid = 0.5 * (a + b)
        nmain = sum(1 for w in lst if w[2] <= mid < w[3]); nres = sum(1 for w in lst if w[1] <= mid < w[4])
        occ[min(nmain, 3)] += b - a; span += b - a
    for w in lst:
        alone = both = pe = 0.0
        for a, b in zip(ev[:-1], ev[1:]):
            if a < w[2] or b > w[3]: continue
            mid = 0.5 * (a + b)
            mates = [m for m in lst if m is not w and m[1] <= mid < m[4]]
            if not mates: alone += b - a
            elif any(m[2] <= mid < m[3] for m in mates): both += b - a
            else: pe += b - a
        A.append([alone, both, pe]); y.append(nph)
A = np.array(A) * T / 1e3; y = np.array(y, dtype=float)
tot = A.sum(axis=0)
print("main-loop time by what the CU-mate does: alone %.1f %%  mate in main loop %.1f %%  mate in prologue / epilogue %.1f %%" % tuple(100 * tot / tot.sum()))
r, *_ = np.linalg.lstsq(A, y, rcond=None)
print("fitted main-loop rate (K-phases of 32 per k tick and workgroup): alone %.3f  with a mate in its main loop %.3f  with a mate in prologue / epilogue %.3f" % tuple(r))
print("  = k ticks per K-phase: alone %.2f  both in main loop %.2f (per CU: %.2f per two phases)  mate in prologue / epilogue %.2f" % (1 / r[0] if r[0] > 0 else float("nan"), 1 / r[1] if r[1] > 0 else float("nan"), 1 / r[1] if r[1] > 0 else float("nan"), 1 / r[2] if r[2] > 0 else float("nan")))
print("CU time with 0 / 1 / 2 workgroups in their main loops: %.1f %% / %.1f %% / %.1f %%" % tuple(100 * occ[k] / span for k in (0, 1, 2)))
