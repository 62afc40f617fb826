"""Row a4 pinned by the reference's own code.  tests/golden/halfsample_sse2_ref.npz holds outputs of the reference's halfSampleSSE2
(reference src/utils.cpp:324-349) executed in the build container (tests/golden/make_halfsample_ref.py through oracle/_ref); the oracle's
SSE2 / REF_X86 modes and the HIP kernel's must reproduce those bytes.  Where oracle/_ref/libdr3ref_halfsample.so exists (the build
container, or a box the snapshot carried it to) the oracle is also held against the live reference function on fresh random planes."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "halfsample_sse2_ref.npz")


def _cases():
    z = np.load(FIX)
    return [(k[3:], z[k], z["ref_" + k[3:]]) for k in z.files if k.startswith("in_")]


def test_fixture_covers_the_widths_the_reference_runs_sse2_on():
    names = [n for n, _, _ in _cases()]
    assert len(names) == 6 and any(n.startswith("hd_1920") for n in names) and "rounding_cases" in names
    for _, img, ref in _cases():
        assert img.shape[1] % 16 == 0 and ref.shape == (img.shape[0] // 2, img.shape[1] // 2) and ref.dtype == np.uint8


def test_oracle_sse2_modes_equal_the_reference_outputs(O):
    for name, img, ref in _cases():
        assert np.array_equal(O.halfsample(img, O.HALF_SSE2), ref), name
        assert np.array_equal(O.halfsample(img, O.HALF_REF_X86), ref), name          # cols % 16 == 0: the path reduce_to_half takes
    # and the double rounding average really differs from the scalar path's truncation on these inputs (Appendix B-2)
    _, img, ref = [c for c in _cases() if c[0] == "rounding_cases"][0]
    assert (O.halfsample(img, O.HALF_TRUNC) != ref).mean() > 0.3


def test_oracle_equals_the_live_reference_function(O):
    lib = os.path.join(ROOT, "oracle", "_ref", "libdr3ref_halfsample.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref is built in the build container only (make -C oracle ref)")
    L = C.CDLL(lib)
    L.dr3ref_halfsample_sse2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(11)
    for w, h in ((16, 2), (32, 7), (1920, 10), (496, 33)):
        raw = np.zeros(w * h + 16, np.uint8); off = (-raw.ctypes.data) % 16
        src = raw[off:off + w * h].reshape(h, w); src[:] = rng.integers(0, 256, (h, w), dtype=np.uint8)
        rawd = np.zeros((w // 2) * (h // 2) + 16, np.uint8); offd = (-rawd.ctypes.data) % 16
        dst = rawd[offd:offd + (w // 2) * (h // 2)].reshape(h // 2, w // 2)
        L.dr3ref_halfsample_sse2(src.ctypes.data, dst.ctypes.data, w, h)
        assert np.array_equal(O.halfsample(src, O.HALF_SSE2), dst), (w, h)


@pytest.mark.gpu
def test_hip_halfsample_equals_the_reference_outputs(dr3, ctx):
    for name, img, ref in _cases():
        for mode in (dr3.HALF_SSE2, dr3.HALF_REF_X86):
            got = ctx.create_img_pyramid(img, 2, mode)[1]
            assert np.array_equal(got, ref), (name, mode)


@pytest.mark.gpu
def test_hip_pyramid_of_1920_wide_levels_follows_the_reference_chain(dr3, ctx):
    """a 1920-wide frame takes the SSE2 path on levels 0..3 (1920, 960, 480, 240 are all multiples of 16): chain the fixture's
    level-0 plane through three more levels on the GPU and through the reference-pinned oracle arithmetic"""
    from oracle import oracle as O
    _, img, ref = [c for c in _cases() if c[0].startswith("hd_1920")][0]
    got = ctx.create_img_pyramid(img[:32], 4, dr3.HALF_REF_X86)
    want = O.halfsample_pyramid(img[:32], 4, O.HALF_REF_X86)
    assert np.array_equal(got[1], ref[:16])
    for l in range(1, 4):
        assert np.array_equal(got[l], want[l]), l


# ---- utils::linspace<float>, the feather ramps' building block (row f1), pinned the same way -----------------------------------------
LIN = os.path.join(ROOT, "tests", "golden", "linspace_ref.npz")


def test_oracle_linspace_equals_the_reference_outputs(O):
    z = np.load(LIN)
    assert len(z.files) == 26
    for k in z.files:
        kind, n = k.split("_"); n = int(n)
        a, b = (0.0, 1.0) if kind == "up" else (1.0, 0.0)
        got = O.linspace_f32(a, b, n)
        assert np.array_equal(got.view(np.uint32), z[k].view(np.uint32)), k          # bit for bit, the inf / nan of N = 1 included


def test_feather_ramp_is_built_from_those_vectors(O):
    """Panorama::add_img_to_canvas (src/panorama.cpp:163-177): left ramp | ones | right ramp; the oracle's and the product's host
    function (dr3hip_panorama_ramp) must hold the reference-produced linspace values at both ends"""
    import importlib
    dr3 = importlib.import_module("3dr_amd")
    z = np.load(LIN)
    for width, fw in ((1000, 300), (700, 50), (640, 100)):
        ramp = O.feather_ramp(0, width, fw)
        assert np.array_equal(ramp[:fw].view(np.uint32), z["up_%d" % fw].view(np.uint32))
        assert np.array_equal(ramp[-fw:].view(np.uint32), z["down_%d" % fw].view(np.uint32))
        assert np.all(ramp[fw:-fw] == 1.0)
        assert np.array_equal(dr3.panorama_ramp(0, width, fw).view(np.uint32), ramp.view(np.uint32))


def test_linspace_live_against_the_reference_function(O):
    lib = os.path.join(ROOT, "oracle", "_ref", "libdr3ref_halfsample.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref is built in the build container only (make -C oracle ref)")
    L = C.CDLL(lib)
    L.dr3ref_linspace_f32.argtypes = [C.c_float, C.c_float, C.c_int, C.c_void_p]
    for a, b, n in ((0.0, 1.0, 77), (1.0, 0.0, 123), (-3.5, 9.25, 1024), (0.1, 0.7, 5)):
        o = np.zeros(n, np.float32)
        L.dr3ref_linspace_f32(a, b, n, o.ctypes.data)
        assert np.array_equal(O.linspace_f32(a, b, n).view(np.uint32), o.view(np.uint32)), (a, b, n)
