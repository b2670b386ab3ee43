"""Generates tests/golden/crazyhorse_features.bin (run in the build container, where /root/reference exists):
decodes the reference's dataset/crazyhorse/*.JPG with PIL into binary PPM files, writes K.txt and runs this repo's own
feature extractor (sfm_opencv_amd/host/NViewReconstruct --features-only: from-scratch SIFT, sfm_features.hpp) on them.
The fixture holds key points, integer-valued SIFT descriptors (one byte per value) and BGR colours of the strongest
1200 key points of each of the 7 images -- derived data, no reference source.

K: the JPEGs carry EXIF FocalLengthIn35mmFilm = 28 (Panasonic DMC-TS3) and were resized to 1024 x 768:
fx = fy = 28 / 36 * 1024 = 796.4, principal point at the image centre.  (The reference hard-codes a K that fits its
3648 x 2736 desktop / dog images, NViewReconstuct.cpp:1353-1356, and notes the TODO itself.)

    python tests/golden/make_crazyhorse_features.py [/root/reference/dataset/crazyhorse] [max key points per image]
"""
import os
import subprocess
import sys
import tempfile

from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/dataset/crazyhorse"
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
out = os.path.join(ROOT, "tests", "golden", "crazyhorse_features.bin")
host = os.path.join(ROOT, "sfm_opencv_amd", "host")
subprocess.check_call(["make", "-C", host, "NViewReconstruct"], stdout=subprocess.DEVNULL)
with tempfile.TemporaryDirectory() as d:
    names = sorted(n for n in os.listdir(src) if n.lower().endswith((".jpg", ".jpeg")))
    for n in names:
        im = Image.open(os.path.join(src, n)).convert("RGB")
        with open(os.path.join(d, os.path.splitext(n)[0] + ".ppm"), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % im.size)
            f.write(im.tobytes())
    w, h = im.size
    with open(os.path.join(d, "K.txt"), "w") as f:
        f.write("%.6f %.6f %.1f %.1f\n" % (28.0 / 36.0 * w, 28.0 / 36.0 * w, w / 2.0, h / 2.0))
    subprocess.check_call([os.path.join(host, "NViewReconstruct"), d, d, "--features-only", "--max-features=%d" % nmax, "--save-features=" + out])
print("wrote", out, os.path.getsize(out), "bytes")
