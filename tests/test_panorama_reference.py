"""moge_amd/panorama.py against the REFERENCE's moge/utils/panorama.py (reference panorama.py:19-50, 109-191).

Two layers:
  * everywhere (this container, the GPU box): against tests/golden/panorama_ref.npz = outputs of the unmodified reference module, written by
    oracle/make_panorama_golden.py (cv2 / utils3d replaced by the stubs documented there);
  * where /root/reference exists: the reference is run LIVE on the same inputs and must reproduce that file (the golden is not stale) and
    agree with moge_amd.panorama.

Tolerances: merged log-distance 5e-5 (two independent constructions of the same sparse system, float32 reference vs float64 operators here,
lsmr stopped at atol = btol = 1e-5 - observed 1.2e-5 / 1.4e-6 / 2.8e-6 at 128 / 256 / 512 pixels), masks equal, uint8 split +-1 LSB (observed 0-1), float split 1e-3 of 255."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import make_panorama_golden as G          # noqa: E402  (test infrastructure: the seeded inputs + the stubs)
from moge_amd import panorama as P        # noqa: E402

GOLD = np.load(G.GOLDEN)
HAVE_REF = os.path.isdir(os.path.join(G.REFERENCE_ROOT, "moge"))
LOG_TOL = 5e-5


def test_cameras_match_the_reference():
    E, Ks = P.get_panorama_cameras()
    assert E.dtype == np.float32 and np.allclose(E, GOLD["cam_E"], atol=1e-6)
    assert np.allclose(np.stack(Ks), GOLD["cam_K"], atol=1e-6)


@pytest.mark.parametrize("name,width,height,res", G.MERGE_CASES)
def test_merge_matches_the_reference(name, width, height, res):
    E, Ks, dist, masks = G.merge_inputs(P, res, seed=width)
    depth, mask = P.merge_panorama_depth(width, height, dist, masks, E, Ks)
    want, want_mask = GOLD[name + "_depth"], GOLD[name + "_mask"]
    assert depth.shape == want.shape and depth.dtype == np.float32
    assert np.array_equal(mask, want_mask)
    assert 0.5 < mask.mean() < 1.0                              # the case has holes AND coverage (view 9 is fully masked, view 3 half)
    err = np.abs(np.log(depth) - np.log(want)).max()
    print("[panorama %s] max |d log distance| vs reference = %.2e" % (name, err))
    assert err < LOG_TOL, err


def test_split_matches_the_reference():
    img = G.split_input(P)
    E, Ks = P.get_panorama_cameras()
    views = np.stack(P.split_panorama_image(img, E, Ks, 64))
    want = GOLD["split_u8"]
    assert views.dtype == np.uint8 and views.shape == want.shape
    d = np.abs(views.astype(np.int16) - want.astype(np.int16))
    print("[panorama split] uint8 max diff %d, differing %.4f %%" % (d.max(), 100.0 * (d > 0).mean()))
    assert d.max() <= 1
    f = np.stack(P.split_panorama_image(img.astype(np.float32), E[:3], Ks[:3], 32))
    assert f.dtype == np.float32 and np.abs(f - GOLD["split_f32"]).max() < 0.255


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference is not present on this box: the committed golden stands in")
def test_live_reference_reproduces_the_golden_and_agrees():
    ref = G.load_reference()
    live = G.reference_outputs(ref, P)
    for k in GOLD.files:
        a, b = live[k], GOLD[k]
        if a.dtype == np.bool_ or a.dtype == np.uint8:
            assert np.array_equal(a, b), k
        elif k.endswith("_depth"):
            assert np.abs(np.log(a) - np.log(b)).max() < 1e-6, k          # same code, same inputs (BLAS / thread order only)
        else:
            assert np.allclose(a, b, atol=1e-6), k
    # the reference's own spherical maps, directly
    uv = P._uv_grid(17, 31)
    assert np.allclose(ref.spherical_uv_to_directions(uv), P.spherical_uv_to_directions(uv), atol=1e-12)
    d = np.random.default_rng(0).normal(size=(50, 3))
    assert np.allclose(ref.directions_to_spherical_uv(d), P.directions_to_spherical_uv(d), atol=1e-12)
