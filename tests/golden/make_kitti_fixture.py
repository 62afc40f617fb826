"""Packs the two KITTI frames the reference's own two-view test reads (reference tests/test_init.cpp:24-25:
../imgs/kitti0.png and ../imgs/kitti5.png, loaded as 8-bit gray) into tests/golden/kitti_pair.npz, so that the parity
tests can run BASELINE configs[0] on the reference's real input data on the GPU box, where /root/reference does not
exist.  Data only: decoded pixels of two images, no code.  Run here (needs /root/reference and Pillow)."""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/imgs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kitti_pair.npz")

frames = {}
for name in ("kitti0", "kitti5"):
    im = Image.open(os.path.join(SRC, name + ".png"))
    assert im.mode == "L", im.mode                       # cv::imread(path, 0): already single-channel gray
    frames[name] = np.asarray(im, np.uint8)
np.savez_compressed(OUT, **frames)
print(OUT, {k: v.shape for k, v in frames.items()}, os.path.getsize(OUT), "bytes")
