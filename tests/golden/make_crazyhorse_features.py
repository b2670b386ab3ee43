"""Generates tests/golden/crazyhorse_features.bin (run in the build container, where /root/reference exists):
links the reference's dataset/crazyhorse/*.JPG into a scratch directory, writes K.txt beside them and runs this repo's own
driver on it (sfm_opencv_amd/host/NViewReconstruct --features-only: baseline JPEG decoder sfm_jpeg.hpp -- bit-identical to
libjpeg's output, tests/test_jpeg_cpu.py -- and the from-scratch SIFT of sfm_features.hpp).
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

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/dataset/crazyhorse"
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
out = os.path.join(ROOT, "tests", "golden", "crazyhorse_features.bin")
host = os.path.join(ROOT, "sfm_opencv_amd", "host")
subprocess.check_call(["make", "-C", host, "NViewReconstruct"], stdout=subprocess.DEVNULL)
w, h = 1024, 768
with tempfile.TemporaryDirectory() as d:
    for n in sorted(os.listdir(src)):
        if n.lower().endswith((".jpg", ".jpeg")):
            os.symlink(os.path.join(src, n), os.path.join(d, n))
    with open(os.path.join(d, "K.txt"), "w") as f:
        f.write("%.6f %.6f %.1f %.1f\n" % (28.0 / 36.0 * w, 28.0 / 36.0 * w, w / 2.0, h / 2.0))
    subprocess.check_call([os.path.join(host, "NViewReconstruct"), d, d, "--sift", "--features-only", "--max-features=%d" % nmax, "--save-features=" + out])
    print("wrote", out, os.path.getsize(out), "bytes")
    # the reference's live configuration: AKAZE key points + 61-byte M-LDB rows (sfm_akaze.hpp), matched under NORM_HAMMING2
    out = os.path.join(ROOT, "tests", "golden", "crazyhorse_features_akaze.bin")
    subprocess.check_call([os.path.join(host, "NViewReconstruct"), d, d, "--akaze", "--features-only", "--max-features=%d" % nmax, "--save-features=" + out])
    print("wrote", out, os.path.getsize(out), "bytes")
