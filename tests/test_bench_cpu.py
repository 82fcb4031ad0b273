"""CPU checks of the measurement definitions in bench.py (no GPU): the algorithmic-byte model must be the one
SURVEY.md section 8(d) states, because roofline.achieved is computed from it."""
import importlib.util
import os

import numpy as np


def _bench():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    spec = importlib.util.spec_from_file_location("bench_module", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey_8d():
    b = _bench()
    N = b.octave_pixels(1920, 1080, 5)
    assert N == [2073600, 518400, 129600, 32400, 8040]                      # sum = 2 762 040 px
    alg = b.algorithmic_bytes_per_frame()
    assert alg["lowpass"] == 16588800                                        # 16.59 MB
    assert abs(alg["scaledown"] - 13.77e6) < 0.01e6                          # 13.77 MB
    assert abs(alg["laplace"] - 88.39e6) < 0.01e6 and abs(alg["detect"] - 77.34e6) < 0.01e6
    total = alg["lowpass"] + alg["scaledown"] + alg["laplace"] + alg["detect"] + 576 * 2000
    assert abs(total - 197.2e6) < 0.1e6                                      # the 197.2 MB/frame of SURVEY 8(d)
    assert alg["dog_scan"] == alg["laplace"] + alg["detect"]                 # the fused kernel is priced as both
    N2 = b.octave_pixels(1280, 960, 5)
    assert N2 == [1228800, 307200, 76800, 19200, 4800]


def test_default_arguments_finish_quickly():
    b = _bench()
    src = open(b.__file__).read()
    assert '"--gpus"' in src and '"--steps"' in src and '"--warmup"' in src    # the driver's contract


def test_dog_scan_flop_model():
    """roofline.frac of the dominant kernel is (146 flop/px x pixels) / time / 157.3 TF: the count is derived in
    code from the blur structure and must stay what DESIGN.md documents."""
    b = _bench()
    assert b.dog_scan_flop_per_px() == 146
    N = b.octave_pixels(1920, 1080, 5)
    assert abs(146 * sum(N) * 64 - 25.8e9) < 0.05e9                         # 25.8 GFLOP per 64-frame step
    assert b.FLOOR_BYTES_PER_FRAME < 54.2e6 < b.ALG_BYTES_PER_FRAME          # floor < measured r01 traffic < algorithmic
