"""Image output of the command-line renderer (RayTracingKit/Image.swift:86-117)."""
import os

import numpy as np


def ppm_text(rgb8):
    """Image.writePPM(to:), Image.swift:86-97: plain PPM (P3), one pixel per line."""
    rgb8 = np.asarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    lines = ["P3", "%d %d" % (w, h), "255"]
    lines += ["%d %d %d" % (p[0], p[1], p[2]) for p in rgb8.reshape(-1, 3)]
    return "\n".join(lines) + "\n"


def write_ppm(path, rgb8):
    """Image.writePPM(to url:), Image.swift:99-117: creates the directory, then writes."""
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    with open(path, "w") as f:
        f.write(ppm_text(rgb8))


def read_ppm(path):
    return read_ppm_text(open(path).read())


def read_ppm_text(text):
    tok = text.split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int64).reshape(h, w, 3).astype(np.uint8)
