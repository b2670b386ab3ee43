"""Generates tests/golden/jpeg/*.jpg and the pixels an independent libjpeg build (Pillow: libjpeg-turbo, default settings =
accurate integer IDCT + fancy upsampling, what cv::imread uses) decodes from them, as binary PPM / PGM.  Run in the build
container (needs Pillow); the fixtures are committed so that tests/test_jpeg_cpu.py runs anywhere.

    python tests/golden/make_jpeg_fixtures.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "jpeg")
os.makedirs(OUT, exist_ok=True)


def scene(h, w, seed):
    """smooth gradients + edges + texture, so that every AC band and the chroma filters are exercised"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(x / 7.0 + y / 13.0), 127 + 120 * np.cos(x / 5.0 - y / 9.0), 255.0 * ((x // 6 + y // 4) % 2)], 2)
    img += rng.normal(0, 25, img.shape)
    img[h // 3: h // 3 + 5, :, :] = (250, 10, 30)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [  # name, size (h, w), mode, save options
    ("c444_q90", (53, 75), "RGB", dict(quality=90, subsampling=0)),
    ("c422_q85", (64, 96), "RGB", dict(quality=85, subsampling=1)),
    ("c422_odd_q60", (45, 67), "RGB", dict(quality=60, subsampling=1)),
    ("c420_q75", (48, 80), "RGB", dict(quality=75, subsampling=2)),
    ("c420_odd_q95", (51, 77), "RGB", dict(quality=95, subsampling=2)),
    ("c420_restart_q80", (70, 90), "RGB", dict(quality=80, subsampling=2, restart_marker_blocks=3)),
    ("c422_restart_rows_opt", (40, 100), "RGB", dict(quality=70, subsampling=1, restart_marker_rows=1, optimize=True)),
    ("gray_q80", (37, 59), "L", dict(quality=80)),
    ("c420_q20", (56, 72), "RGB", dict(quality=20, subsampling=2)),
]
for k, (name, (h, w), mode, opts) in enumerate(CASES):
    a = scene(h, w, 100 + k)
    im = Image.fromarray(a).convert(mode)
    path = os.path.join(OUT, name + ".jpg")
    im.save(path, "JPEG", **opts)
    dec = Image.open(path)
    dec.load()
    ext = ".pgm" if mode == "L" else ".ppm"
    dec.save(os.path.join(OUT, name + ext))
    print(name, os.path.getsize(path), "bytes ->", dec.size, dec.mode)
# one progressive file: the decoder must refuse it (no pixels)
Image.fromarray(scene(40, 40, 7)).save(os.path.join(OUT, "progressive_unsupported.jpg"), "JPEG", quality=80, progressive=True)
