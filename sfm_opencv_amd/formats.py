"""structure.yml / binary .ply: the on-disk contract of the reference.

save_structure (NViewReconstuct.cpp:186-227) writes through cv::FileStorage (OpenCV 4.4 YAML 1.0 emitter [3P]):
    %YAML:1.0 / --- / "Camera Count" / "Point Count" / Rotations, Motions = sequences of !!opencv-matrix maps /
    Points = block sequence of flow sequences [x, y, z] / Colors = [b, g, r]
    doubles "%.16e", integral values as "1." ; flow sequences wrap when offset + token would pass column 71.
write_ply_binary (NView:229-294): text header + packed little-endian <6f3B vertices, rows containing NaN skipped.
get_ply_pts3d (NView:296-338): float32 casts, BGR -> RGB.
These byte-reproduce the reference's own files in tests/golden (tests/test_formats.py).
"""
import re
from decimal import Decimal, ROUND_HALF_UP

import numpy as np

_WRAP = 71      # cv::FileStorage wrap margin [3P]

PLY_VERTEX = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
                       ("r", "u1"), ("g", "u1"), ("b", "u1")])


def _dtoa(v):
    """fs::doubleToString(buf, value, explicitZero=false) [3P]"""
    v = float(v)
    if np.isnan(v):
        return ".Nan"
    if np.isinf(v):
        return "-.Inf" if v < 0 else ".Inf"
    if abs(v) < 2 ** 31 and float(int(round(v))) == v:      # cvRound(value) == value
        return "%d." % int(round(v))
    # sprintf("%.16e") of the Windows CRT the reference ran on: exact decimal ties round half AWAY from zero
    # (golden files: 1.85865020751953125 -> "1.8586502075195313e+00"; glibc would print ...312).
    d = Decimal(v)
    exp10 = d.adjusted()
    q = d.scaleb(-exp10).quantize(Decimal(1).scaleb(-16), rounding=ROUND_HALF_UP)
    if abs(q) >= 10:
        exp10 += 1
        q = d.scaleb(-exp10).quantize(Decimal(1).scaleb(-16), rounding=ROUND_HALF_UP)
    return "%se%s%02d" % (format(q, "f"), "+" if exp10 >= 0 else "-", abs(exp10))


def _ftoa(v):
    """fs::floatToString(buf, value, halfprecision=false, explicitZero=false) [3P]: "%.8e" of a float32"""
    v = float(np.float32(v))
    if np.isnan(v):
        return ".Nan"
    if np.isinf(v):
        return "-.Inf" if v < 0 else ".Inf"
    if abs(v) < 2 ** 31 and float(int(round(v))) == v:
        return "%d." % int(round(v))
    d = Decimal(v)
    exp10 = d.adjusted()
    q = d.scaleb(-exp10).quantize(Decimal(1).scaleb(-8), rounding=ROUND_HALF_UP)
    if abs(q) >= 10:
        exp10 += 1
        q = d.scaleb(-exp10).quantize(Decimal(1).scaleb(-8), rounding=ROUND_HALF_UP)
    return "%se%s%02d" % (format(q, "f"), "+" if exp10 >= 0 else "-", abs(exp10))


class _Emitter:
    def __init__(self):
        self.lines = []
        self.cur = ""

    def flush(self, indent):
        self.lines.append(self.cur)
        self.cur = " " * indent

    def flow_seq(self, prefix, tokens, indent):
        """`prefix[ t0, t1, ... ]`, wrapping like YAMLEmitter::writeScalar in a flow collection."""
        self.cur = prefix + "["
        first = True
        for t in tokens:
            if not first:
                self.cur += ","
            if len(self.cur) + len(t) > _WRAP and len(self.cur) > indent:
                self.flush(indent)
            else:
                self.cur += " "
            self.cur += t
            first = False
        self.cur += " ]"
        self.lines.append(self.cur)
        self.cur = ""


def structure_yml_text(rotations, motions, points, colors):
    """Text of save_structure's output. rotations: list of 3x3, motions: list of 3x1, points (n,3) double,
    colors (n,3) uint8 in the order given (the reference passes BGR)."""
    e = _Emitter()
    e.lines += ["%YAML:1.0", "---", "Camera Count: %d" % len(rotations), "Point Count: %d" % len(points)]

    def mats(name, ms, rows, cols):
        e.lines.append(name + ":")
        for m in ms:
            m = np.asarray(m, np.float64).reshape(rows, cols)
            e.lines += ["   - !!opencv-matrix", "      rows: %d" % rows, "      cols: %d" % cols, "      dt: d"]
            e.flow_seq("      data: ", [_dtoa(v) for v in m.reshape(-1)], 10)

    mats("Rotations", rotations, 3, 3)
    mats("Motions", motions, 3, 1)
    e.lines.append("Points:")
    for p in np.asarray(points, np.float64).reshape(-1, 3):
        e.flow_seq("   - ", [_dtoa(v) for v in p], 7)
    e.lines.append("Colors:")
    for c in np.asarray(colors).reshape(-1, 3):
        e.flow_seq("   - ", ["%d" % int(v) for v in c], 7)
    return "\n".join(e.lines) + "\n"


def structure_yml_text_twoview(rotations, motions, structure_h, colors):
    """TwoViewReconstruct.cpp:313-356: `structure_h` is the homogeneous 4 x N float32 matrix of cv::triangulatePoints; every
    column is divided by its w in float32 (Mat_<float> c; c /= c(3): product with the double reciprocal, rounded to float
    [3P]) and written as a Point3f ("%.8e")."""
    e = _Emitter()
    h = np.asarray(structure_h, np.float32).reshape(4, -1)
    e.lines += ["%YAML:1.0", "---", "Camera Count: %d" % len(rotations), "Point Count: %d" % h.shape[1]]
    for name, ms, rows, cols in (("Rotations", rotations, 3, 3), ("Motions", motions, 3, 1)):
        e.lines.append(name + ":")
        for m in ms:
            m = np.asarray(m, np.float64).reshape(rows, cols)
            e.lines += ["   - !!opencv-matrix", "      rows: %d" % rows, "      cols: %d" % cols, "      dt: d"]
            e.flow_seq("      data: ", [_dtoa(v) for v in m.reshape(-1)], 10)
    e.lines.append("Points:")
    rw = 1.0 / h[3].astype(np.float64)
    xyz = (h[:3].astype(np.float64) * rw).astype(np.float32)
    for i in range(h.shape[1]):
        e.flow_seq("   - ", [_ftoa(v) for v in xyz[:, i]], 7)
    e.lines.append("Colors:")
    for c in np.asarray(colors).reshape(-1, 3):
        e.flow_seq("   - ", ["%d" % int(v) for v in c], 7)
    return "\n".join(e.lines) + "\n"


def save_structure(file_name, rotations, motions, structure, colors):
    """NViewReconstuct.cpp:186-227."""
    with open(file_name, "w", newline="\n") as f:
        f.write(structure_yml_text(rotations, motions, structure, colors))


def read_structure_yml(path):
    txt = open(path, "r").read()
    ncam = int(re.search(r"Camera Count:\s*(\d+)", txt).group(1))
    npt = int(re.search(r"Point Count:\s*(\d+)", txt).group(1))
    sec = {}
    names = ["Rotations", "Motions", "Points", "Colors"]
    pos = [txt.index("\n" + n + ":") for n in names] + [len(txt)]
    for i, n in enumerate(names):
        sec[n] = txt[pos[i]:pos[i + 1]]

    def num(s):
        s = s.strip()
        return {".Nan": np.nan, ".Inf": np.inf, "-.Inf": -np.inf}.get(s, None) if s in (".Nan", ".Inf", "-.Inf") else float(s)

    def seqs(s):
        return [[num(t) for t in m.split(",")] for m in re.findall(r"\[([^\]]*)\]", s)]

    rot = [np.array(v).reshape(3, 3) for v in seqs(sec["Rotations"])]
    mot = [np.array(v).reshape(3, 1) for v in seqs(sec["Motions"])]
    pts = np.array(seqs(sec["Points"]), np.float64).reshape(-1, 3)
    col = np.array(seqs(sec["Colors"]), np.float64).reshape(-1, 3).astype(np.uint8)
    assert len(rot) == ncam and len(mot) == ncam and pts.shape[0] == npt
    return dict(rotations=rot, motions=mot, points=pts, colors=col)


def get_ply_pts3d(pts3d, normals, colors):
    """NViewReconstuct.cpp:296-338: float32 casts, colors[i] = (b, g, r) -> r, g, b.  Returns (ret, vertices)."""
    pts3d = np.asarray(pts3d, np.float64).reshape(-1, 3); normals = np.asarray(normals, np.float64).reshape(-1, 3)
    colors = np.asarray(colors).reshape(-1, 3)
    if not (len(pts3d) == len(normals) == len(colors)):
        print("[Err]: items size not equal.")
        return -1, np.zeros(0, PLY_VERTEX)
    v = np.zeros(len(pts3d), PLY_VERTEX)
    v["x"], v["y"], v["z"] = pts3d[:, 0], pts3d[:, 1], pts3d[:, 2]
    v["nx"], v["ny"], v["nz"] = normals[:, 0], normals[:, 1], normals[:, 2]
    v["b"], v["g"], v["r"] = colors[:, 0], colors[:, 1], colors[:, 2]
    print("Total %d 3D points." % len(pts3d))
    return 0, v


def ply_bytes(vertices, newline="\n"):
    """write_ply_binary's bytes (NView:229-294).  The reference's header has CRLF because it was written in
    Windows text mode; newline="\\r\\n" reproduces its files byte for byte, the default emits LF."""
    v = np.asarray(vertices, PLY_VERTEX)
    bad = np.zeros(len(v), bool)
    for k in ("x", "y", "z", "nx", "ny", "nz"):
        bad |= np.isnan(v[k])
    v = v[~bad]
    hdr = ["ply", "format binary_little_endian 1.0", "element vertex %d" % len(v),
           "property float x", "property float y", "property float z",
           "property float nx", "property float ny", "property float nz",
           "property uchar red", "property uchar green", "property uchar blue", "end_header"]
    return (newline.join(hdr) + newline).encode("ascii") + v.tobytes()


def write_ply_binary(path, vertices, newline="\n"):
    with open(path, "wb") as f:
        f.write(ply_bytes(vertices, newline))


def read_ply_binary(path):
    raw = open(path, "rb").read()
    end = raw.index(b"end_header") + len(b"end_header")
    while raw[end:end + 1] in (b"\r", b"\n"):
        end += 1
        if raw[end - 1:end] == b"\n":
            break
    n = int(re.search(rb"element vertex (\d+)", raw[:end]).group(1))
    return np.frombuffer(raw[end:end + n * PLY_VERTEX.itemsize], PLY_VERTEX).copy()
