"""Binary PGM (P5, 8-bit) reader for the grey test frames under tests/golden/."""
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_pgm(path):
    with open(path, "rb") as f:
        data = f.read()
    parts = data.split(None, 4)
    if parts[0] != b"P5" or int(parts[3]) != 255:
        raise ValueError(f"{path}: not an 8-bit binary PGM")
    w, h = int(parts[1]), int(parts[2])
    return np.frombuffer(data, np.uint8, w * h, len(data) - w * h).reshape(h, w).copy()


def golden_frame(name):
    return read_pgm(os.path.join(_ROOT, "tests", "golden", name))
