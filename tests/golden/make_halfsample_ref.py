"""Makes tests/golden/halfsample_sse2_ref.npz and tests/golden/linspace_ref.npz: outputs of the REFERENCE'S OWN halfSampleSSE2 (reference src/utils.cpp:324-349), executed
in the build container through oracle/_ref/libdr3ref_halfsample.so (`make -C oracle ref` compiles the function from where it lies under
/root/reference; nothing of its source is stored here).  The fixture is data: input planes (seeded random + every rounding case of the
two averaging stages) and the bytes the reference produced.  It pins row a4's SSE2 arithmetic -- the path utils::reduce_to_half takes
when cols % 16 == 0 and both buffers are 16-byte aligned (src/utils.cpp:387-392): 1920-wide levels 0..3, never 1240 / 1241.
usage (build container only): python tests/golden/make_halfsample_ref.py"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def aligned(shape, align=16):
    n = int(np.prod(shape))
    raw = np.zeros(n + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].reshape(shape)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdr3ref_halfsample.so"))
    L.dr3ref_halfsample_sse2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.dr3ref_halfsample_sse2.restype = None
    rng = np.random.default_rng(0x3D52)
    planes = {}
    # the widths the reference's SSE2 path can see: 1920 (configs[4], level 0), its next levels 960 / 480 / 240, and 1248 (a /16 width
    # near KITTI's); heights even and odd (the function drops the last row of an odd height: h >> 1)
    for name, (w, h) in {"hd_1920x38": (1920, 38), "l1_960x27": (960, 27), "l2_480x14": (480, 14), "l3_240x9": (240, 9), "k_1248x24": (1248, 24)}.items():
        planes[name] = rng.integers(0, 256, (h, w), dtype=np.uint8)
    # every rounding case: a 2x2 block (a b / c d) -> avg16(avg8(a,c), avg8(b,d)); all 4-tuples over a set of values that exercises both
    # round-half-up stages and the 255 end
    vals = np.array([0, 1, 2, 3, 4, 127, 128, 129, 253, 254, 255], np.uint8)
    g = np.stack(np.meshgrid(vals, vals, vals, vals, indexing="ij"), -1).reshape(-1, 4)        # 14641 blocks
    nb = -(-len(g) // 8) * 8
    g = np.concatenate([g, np.zeros((nb - len(g), 4), np.uint8)])
    cases = np.zeros((2, 2 * nb), np.uint8)
    cases[0, 0::2], cases[0, 1::2], cases[1, 0::2], cases[1, 1::2] = g[:, 0], g[:, 1], g[:, 2], g[:, 3]
    planes["rounding_cases"] = cases
    out = {}
    for name, img in planes.items():
        h, w = img.shape
        assert w % 16 == 0
        src = aligned((h, w)); src[:] = img
        dst = aligned((h // 2, w // 2)); dst[:] = 0xA5
        L.dr3ref_halfsample_sse2(src.ctypes.data, dst.ctypes.data, w, h)
        out["in_" + name] = img
        out["ref_" + name] = dst.copy()
    np.savez_compressed(os.path.join(HERE, "halfsample_sse2_ref.npz"), **out)
    print("wrote halfsample_sse2_ref.npz:", {k: v.shape for k, v in out.items() if k.startswith("ref_")})
    # utils::linspace<float> (reference include/utils.hpp:36-46): the feather ramps of Panorama::add_img_to_canvas are
    # linspace(0, 1, w) and linspace(1, 0, w) (src/panorama.cpp:171-172); the accumulating `val += h` is what makes them differ from
    # a closed form in the last bits.  Widths: the reference's feathering widths and the degenerate ones (N = 1 divides by zero: inf / nan)
    L.dr3ref_linspace_f32.argtypes = [C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.dr3ref_linspace_f32.restype = None
    lin = {}
    for n in (1, 2, 3, 7, 50, 51, 100, 299, 300, 301, 640, 1000, 4000):
        for name, (a, b) in (("up", (0.0, 1.0)), ("down", (1.0, 0.0))):
            o = np.zeros(n, np.float32)
            L.dr3ref_linspace_f32(a, b, n, o.ctypes.data)
            lin["%s_%d" % (name, n)] = o
    np.savez_compressed(os.path.join(HERE, "linspace_ref.npz"), **lin)
    print("wrote linspace_ref.npz:", len(lin), "vectors")


if __name__ == "__main__":
    main()
