#!/usr/bin/env python3
"""Cuts the test input images out of the only real images in the reference tree
(Thirdparty/libelas-gpu/input/*.pgm) and stores them under
tests/golden/ as binary PGM.  Dev-time tool: needs /root/reference; the tests
and the GPU box only ever read the committed crops."""
import os

import numpy as np

SRC = "/root/reference/Thirdparty/libelas-gpu/input"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def read_pgm(path):
    with open(path, "rb") as f:
        data = f.read()
    toks, pos = [], 0
    while len(toks) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        toks.append(data[pos:end])
        pos = end
    assert toks[0] == b"P5" and int(toks[3]) == 255
    w, h = int(toks[1]), int(toks[2])
    return np.frombuffer(data, np.uint8, w * h, pos + 1).reshape(h, w)


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img).tobytes())


if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    crops = [("aloe_left", 640, 480, 64, 32, "aloe_640x480"), ("aloe_left", 640, 480, 67, 34, "aloe_640x480_shift"),
             ("cones_left", 640, 480, 128, 96, "cones_640x480"), ("urban1_left", 1241, 376, 40, 8, "urban1_1241x376"),
             ("urban1_right", 1241, 376, 40, 8, "urban1_right_1241x376")]  # rectified pair for M5
    for name, w, h, x0, y0, out in crops:
        img = read_pgm(os.path.join(SRC, name + ".pgm"))
        write_pgm(os.path.join(DST, out + ".pgm"), img[y0:y0 + h, x0:x0 + w])
        print(out, img.shape)
