"""Generate the committed fixtures under tests/golden/ (run HERE, where /root/reference exists).

  stereo_pair_u8.npz   the reference's only sample images, data/left.pgm and data/righ.pgm
                       (1280x960, 8-bit P5), stored as compressed uint8 arrays 'left', 'right'.
                       /root/reference does not exist on the GPU box, so tests read this file.
  oracle_small.npz     oracle outputs on a 320x240 crop of left.pgm (regression pin of the oracle
                       itself: counters, sorted keypoint fields, descriptor checksum).

  refemul_golden.npz   GOLDEN VECTORS FROM THE REFERENCE ITSELF: outputs of the reference's own kernels and host code
                       (cudaSiftH.cu + cudaSiftD.cu + matching.cu, compiled by oracle/build_ref.sh against the CPU SIMT
                       emulator, -ffp-contract=fast flavour) — sha256 of the dense stages on the crop, the complete
                       ExtractSift records of the crop and of left.pgm, MatchSiftData and FindHomography results on
                       seeded synthetic inputs, and the numbers the reference's demo program prints.  They travel to
                       the GPU box, where /root/reference (and possibly oracle/_ref) does not exist.

Usage:  python tests/golden/make_fixtures.py
"""
import hashlib
import re
import subprocess
import tempfile
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
    refemul_golden(left, right, crop)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


DESC_STRIDE = 4


def wide_1080p(left_u8):
    """1920x1080 from the 1280x960 left.pgm by mirroring 320 columns / 60 rows outwards (np.pad reflect)."""
    return np.pad(left_u8, ((60, 60), (320, 320)), mode="reflect").astype(np.float32)


def sorted_records(pts, total):
    p = pts[:total]
    order = np.lexsort((p["orientation"], p["scale"], p["xpos"], p["ypos"], p["subsampling"]))
    return p[order]


def refemul_golden(left, right, crop):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyrefemul as ref
    from synth import synth_descriptors, descriptors_to_points, synth_matches
    from oracle.pyoracle import POINT_DTYPE
    assert ref.available("fast"), "build oracle/_ref first (make -C oracle ref)"
    out = {}
    # dense stages on the crop (bit patterns)
    low = ref.lowpass(crop, 1.0, "fast")
    out["sha_lowpass"] = sha(low)
    out["sha_scaledown"] = sha(ref.scaledown(low, "fast"))
    out["sha_laplace"] = sha(ref.laplace(low, 5, 5, "fast"))
    odd = crop[:37, :131].copy()                              # ragged sizes: clamp paths, width % 4 != 0
    out["sha_lowpass_odd"] = sha(ref.lowpass(odd, 1.3, "fast"))
    out["sha_scaledown_odd"] = sha(ref.scaledown(odd, "fast"))
    out["sha_laplace_odd"] = sha(ref.laplace(odd, 5, 3, "fast"))
    # whole ExtractSift
    # "wide" = the bench workload's shape and parameters (1920x1080, 5 octaves, initBlur 1.0, thresh 3.0: mainSift.cpp:58-67)
    # on a natural image: left.pgm mirrored outwards (integer indexing only, so the input is reproducible from the
    # committed stereo pair); "righ" = the other image of the pair at the demo's threshold; "crop_up" = scaleUp
    for name, img, noct, th, up in (("crop", crop, 4, 3.5, False), ("left", left.astype(np.float32), 5, 4.5, False),
                                    ("wide", wide_1080p(left), 5, 3.0, False), ("righ", right.astype(np.float32), 5, 4.5, False),
                                    ("crop_up", crop, 4, 3.5, True)):
        pts, n, cnt = ref.extract(img, noct, 1.0, th, scale_up=up, flavour="fast")
        total = int(cnt[2 * noct + 1])
        out[name + "_n"] = n
        out[name + "_counters"] = cnt
        recs = sorted_records(pts, total)
        if name in ("wide", "righ", "crop_up"):        # keep the file small: every field of every record, but the 128-float
            recs["data"][np.arange(total) % DESC_STRIDE != 0] = 0.0     # descriptor of every DESC_STRIDE-th (sorted) record only
        out[name + "_records"] = recs
    # MatchSiftData on seeded descriptors (n2 % 32 != 0: the reference ignores the last n2 % 32 columns)
    a = descriptors_to_points(synth_descriptors(1000, 7), POINT_DTYPE)
    b = descriptors_to_points(synth_descriptors(1500, 8), POINT_DTYPE)
    ref.match(a, 1000, b, 1500, "fast")
    for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
        out["match_" + f] = a[f].copy()
    # FindHomography (libc rand() seeded with 1, the reference draws its samples from it)
    m, _, _ = synth_matches(3000, seed=5, dtype=POINT_DTYPE)
    H, nm = ref.find_homography(m, 3000, 2000, 0.85, 0.95, 5.0, seed=1, flavour="fast")
    out["homography_H"] = H
    out["homography_inliers"] = nm
    # the reference's demo program (mainSift.cpp + geomFuncs.cpp) on the emulated library
    exe = os.path.join(ROOT, "oracle", "_ref", "cudasift_refemul_main")
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "data"))
        for nm_, img in (("left", left), ("righ", right)):
            with open(os.path.join(tmp, "data", nm_ + ".pgm"), "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
                f.write(img.tobytes())
        r = subprocess.run([exe, "0", "1"], cwd=tmp, capture_output=True, text=True, check=True,
                           env=dict(os.environ, SIMT_THREADS="1"))
    m1 = re.search(r"Number of original features: (\d+) (\d+)", r.stdout)
    m2 = re.search(r"Number of matching features: (\d+) (\d+)", r.stdout)
    out["main_features"] = np.array([int(m1.group(1)), int(m1.group(2))])
    out["main_matching"] = np.array([int(m2.group(1)), int(m2.group(2))])
    np.savez_compressed(os.path.join(HERE, "refemul_golden.npz"), **out)
    print("wrote refemul_golden.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()
                                        if not k.endswith("records")}, "main:", m1.group(0), "|", m2.group(0))


if __name__ == "__main__":
    main()
