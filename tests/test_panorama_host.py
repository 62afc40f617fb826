"""Host-side pieces of the panorama paste against the oracle: Panorama::set_bbox and the feather
ramp of add_img_to_canvas (reference src/panorama.cpp:112-138, :163-177).  No GPU needed: both are plain
host functions of the C ABI."""
import numpy as np
import pytest


def _rand_M(rng):
    a = rng.uniform(-0.2, 0.2)
    return np.array([[np.cos(a) * rng.uniform(.7, 1.4), -np.sin(a), rng.uniform(-50, 400)],
                     [np.sin(a), np.cos(a) * rng.uniform(.7, 1.4), rng.uniform(-50, 300)],
                     [rng.uniform(-2e-4, 2e-4), rng.uniform(-2e-4, 2e-4), 1]], np.float32)


def test_bbox_matches_oracle(dr3, O):
    rng = np.random.default_rng(5)
    for _ in range(200):
        M = _rand_M(rng)
        w, h = int(rng.integers(8, 4000)), int(rng.integers(8, 3000))
        assert dr3.panorama_bbox(M, w, h) == O.panorama_bbox(M, w, h)
    # identity: int() truncation of exact corners
    assert dr3.panorama_bbox(np.eye(3, dtype=np.float32), 752, 480) == (0, 0, 752, 480)
    M = np.array([[1, 0, -3.7], [0, 1, 2.9], [0, 0, 1]], np.float32)        # int() truncates toward zero
    assert dr3.panorama_bbox(M, 100, 50) == (-3, 2, 96, 52) == O.panorama_bbox(M, 100, 50)


def test_ramp_matches_oracle_and_reference_quirks(dr3, O):
    for n in (2, 3, 4, 5, 9, 10, 64, 301, 1000):
        for fw in (0, 1, 2, 5, 50, 150, 2000):
            a, b = dr3.panorama_ramp(7, 7 + n, fw), O.feather_ramp(7, 7 + n, fw)
            assert a.dtype == np.float32 and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (n, fw)
            assert len(a) == n
    r = dr3.panorama_ramp(0, 100, 10)
    assert r[0] == 0.0 and r[-1] != 0.0                 # the accumulating linspace ends one ulp-ish off 0
    assert np.all(r[10:90] == 1.0) and abs(r[9] - 1.0) < 1e-6
    r = dr3.panorama_ramp(0, 10, 50)                    # box narrower than two ramps: temp = n/2 - 1
    assert len(r) == 10 and np.all(r[4:6] == 1.0)
    assert np.array_equal(dr3.panorama_ramp(0, 4, 1), np.array([0, 1, 1, 1], np.float32))   # linspace(N = 1): [a]
    for bad in (0, 1):
        with pytest.raises(dr3.Dr3HipError):
            dr3.panorama_ramp(0, bad, 3)
        with pytest.raises(ValueError):
            O.feather_ramp(0, bad, 3)
