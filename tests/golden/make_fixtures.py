"""Generate the committed fixtures under tests/golden/ (run HERE, where /root/reference exists).

  stereo_pair_u8.npz   the reference's only sample images, data/left.pgm and data/righ.pgm
                       (1280x960, 8-bit P5), stored as compressed uint8 arrays 'left', 'right'.
                       /root/reference does not exist on the GPU box, so tests read this file.
  oracle_small.npz     oracle outputs on a 320x240 crop of left.pgm (regression pin of the oracle
                       itself: counters, sorted keypoint fields, descriptor checksum).

Usage:  python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("REF", "/root/reference")


def read_pgm(path):
    data = open(path, "rb").read()
    toks, i = [], 0
    while len(toks) < 4:
        while data[i:i + 1].isspace():
            i += 1
        if data[i:i + 1] == b"#":
            while data[i:i + 1] != b"\n":
                i += 1
            continue
        j = i
        while not data[j:j + 1].isspace():
            j += 1
        toks.append(data[i:j])
        i = j
    i += 1
    assert toks[0] == b"P5" and int(toks[3]) == 255
    w, h = int(toks[1]), int(toks[2])
    return np.frombuffer(data[i:i + w * h], dtype=np.uint8).reshape(h, w).copy()


def main():
    left = read_pgm(os.path.join(REF, "data", "left.pgm"))
    right = read_pgm(os.path.join(REF, "data", "righ.pgm"))
    np.savez_compressed(os.path.join(HERE, "stereo_pair_u8.npz"), left=left, right=right)
    from oracle import pyoracle as orc
    crop = left[300:540, 400:720].astype(np.float32)
    pts, n, cnt = orc.extract(crop, num_octaves=4, init_blur=1.0, thresh=3.5)
    p = pts[:n]
    order = np.lexsort((p["orientation"], p["scale"], p["xpos"], p["ypos"]))
    p = p[order]
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"), counters=cnt, n=n,
                        xpos=p["xpos"], ypos=p["ypos"], scale=p["scale"], orientation=p["orientation"],
                        sharpness=p["sharpness"], edgeness=p["edgeness"], desc=p["data"])
    print("wrote fixtures: left/right", left.shape, "oracle_small n =", n, cnt[:10])


if __name__ == "__main__":
    main()
