"""Drop-in check: the reference's OWN mainSift.cpp + geomFuncs.cpp (compiled unchanged from the reference
tree by `make dropin` into oracle/_ref/cudasift_dropin, linked against libcudasift.so + libmisift.so) runs
on the GPU and prints what the reference prints (mainSift.cpp:80-81), with the feature counts of the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "cudasift_dropin")
BIN_MANAGED = os.path.join(ROOT, "oracle", "_ref", "cudasift_dropin_managed")


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())


@pytest.mark.parametrize("flavour", ["default", "managed"])
def test_reference_main_runs_unchanged(tmp_path, flavour):
    """default: SiftData = {h_data, d_data}; managed: the reference's -DMANAGEDMEM flavour (cudaSift.h:27-32, one
    hipMallocManaged pointer m_data read by the host code of mainSift.cpp / geomFuncs.cpp directly)."""
    exe = BIN if flavour == "default" else BIN_MANAGED
    if not os.path.exists(exe):
        pytest.skip("%s not built (needs /root/reference at build time)" % exe)
    z = np.load(os.path.join(ROOT, "tests", "golden", "stereo_pair_u8.npz"))
    os.makedirs(tmp_path / "data")
    write_pgm(tmp_path / "data" / "left.pgm", z["left"])
    write_pgm(tmp_path / "data" / "righ.pgm", z["right"])
    env = dict(os.environ, MISIFT_QUIET="1")
    r = subprocess.run([exe, "0", "1"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    outdir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(outdir):
        with open(os.path.join(outdir, "dropin_stdout%s.txt" % ("" if flavour == "default" else "_managed")), "w") as f:
            f.write(out[-6000:])
    assert "Image size = (1280,960)" in out
    m = re.search(r"Number of original features: (\d+) (\d+)", out)
    assert m, out[-2000:]
    from oracle import pyoracle as orc
    _, n1, _ = orc.extract(z["left"].astype(np.float32), 5, 1.0, 4.5)
    _, n2, _ = orc.extract(z["right"].astype(np.float32), 5, 1.0, 4.5)
    assert (int(m.group(1)), int(m.group(2))) == (n1, n2) and n1 > 64 and n2 > 64
    m2 = re.search(r"Number of matching features: (\d+) (\d+)", r.stdout)
    assert m2 and int(m2.group(2)) >= 8, r.stdout[-1500:]            # the 7-px shift is found
    m2 = re.search(r"Number of matching features: (\d+) (\d+) ([\d.]+)% 1 4.5", out)
    assert m2, out[-2000:]
    assert int(m2.group(2)) > 100                       # RANSAC inliers of the best hypothesis
    assert os.path.exists(tmp_path / "data" / "limg_pts.pgm")


def test_reference_main_on_a_100x100_image(tmp_path):
    """mainSift.cpp asks for 5 octaves (mainSift.cpp:59): on a 100 x 100 image the coarsest level is 6 px.  The
    reference runs that (cudaSiftH.cu:72-167) and so must the drop-in — r04 ended the caller's process here."""
    if not os.path.exists(BIN):
        pytest.skip("%s not built (needs /root/reference at build time)" % BIN)
    z = np.load(os.path.join(ROOT, "tests", "golden", "stereo_pair_u8.npz"))
    os.makedirs(tmp_path / "data")
    # two views 7 px apart of the most textured 100 x 100 patch of left.pgm (119 / 107 features at the demo's thresh 4.5).
    # NOT any crop: with fewer than 32 features in the second image MatchSiftData compares nothing (matching.cu:1103:
    # 32 * (numPts2 / 32) columns), every match stays -1 and the reference's OWN PrintMatchData then draws from
    # sift2[-1] (mainSift.cpp:163-184) — the demo segfaults on such a pair with or without this library.
    left, right = z["left"][500:600, 700:800], z["left"][500:600, 707:807]
    write_pgm(tmp_path / "data" / "left.pgm", left)
    write_pgm(tmp_path / "data" / "righ.pgm", right)
    r = subprocess.run([BIN, "0", "1"], cwd=tmp_path, env=dict(os.environ, MISIFT_QUIET="1"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "Image size = (100,100)" in r.stdout
    m = re.search(r"Number of original features: (\d+) (\d+)", r.stdout)
    assert m, r.stdout[-2000:]
    from oracle import pyoracle as orc
    _, n1, _ = orc.extract(left.astype(np.float32), 5, 1.0, 4.5)
    _, n2, _ = orc.extract(right.astype(np.float32), 5, 1.0, 4.5)
    assert (int(m.group(1)), int(m.group(2))) == (n1, n2) and n1 > 64 and n2 > 64
    m2 = re.search(r"Number of matching features: (\d+) (\d+)", r.stdout)
    assert m2 and int(m2.group(2)) >= 8, r.stdout[-1500:]            # the 7-px shift is found
