"""Measured accuracy of the written-out elementary functions shared by oracle/sift_oracle.c and the HIP kernels
(det_exp2 / det_atan2 / det_exp / det_sincos) against float64 libm -> profiles/r03_det_accuracy.json.
The assertions live in tests/test_oracle_cpu.py::test_det_functions_accuracy; this writes the numbers down."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402


def ulp_err(got, want64):
    ulp = np.maximum(np.spacing(np.abs(want64.astype(np.float32))).astype(np.float64), np.finfo(np.float32).tiny)
    return np.abs(got.astype(np.float64) - want64) / ulp


def main():
    rng = np.random.default_rng(42)
    n = 1 << 22
    out = {"inputs_per_function": n, "reference": "numpy float64 (glibc libm)"}
    x = np.concatenate([rng.uniform(-0.12, 0.12, n // 2), rng.uniform(-30, 30, n // 2)]).astype(np.float32)
    e = ulp_err(orc.det_eval(0, x), np.exp2(x.astype(np.float64)))
    out["det_exp2"] = {"range": "[-0.12, 0.12] (pds/5) and [-30, 30]", "max_ulp": float(e.max()), "replaces": "exp2f (cudaSiftD.cu:1417), CUDA documents 2 ulp"}
    gx = np.concatenate([rng.uniform(-255, 255, n // 2), rng.normal(0, 1e-3, n // 2)]).astype(np.float32)
    gy = np.concatenate([rng.uniform(-255, 255, n // 2), rng.normal(0, 1e-3, n // 2)]).astype(np.float32)
    got, want = orc.det_eval(1, gx, gy), np.arctan2(gy.astype(np.float64), gx.astype(np.float64))
    out["det_atan2"] = {"range": "image gradients in [-255, 255]^2 and N(0, 1e-3)^2", "max_abs_rad": float(np.abs(got - want).max()),
                        "max_ulp": float(ulp_err(got, want)[np.abs(want) > 1e-3].max()),
                        "bins_of_32": float(np.abs(got - want).max() * 16 / 3.1416), "replaces": "atan2f (cudaSiftD.cu:1008), CUDA documents 2 ulp"}
    x = np.concatenate([-rng.uniform(0, 1, n // 2), -rng.uniform(0, 80, n // 2)]).astype(np.float32)
    e = ulp_err(orc.det_eval(2, x), np.exp(x.astype(np.float64)))
    out["det_exp"] = {"range": "[-80, 0]", "max_ulp": float(e.max()), "replaces": "exp (cudaSiftD.cu:987, 2 ulp) and __expf (:317, 2 + |x/ln2| ulp)"}
    x = rng.uniform(0.0, 2.0 * 3.1415, n).astype(np.float32)
    s, c = orc.det_eval(3, x)
    out["det_sincos"] = {"range": "[0, 2*3.1415]", "max_abs_sin": float(np.abs(s - np.sin(x.astype(np.float64))).max()),
                         "max_abs_cos": float(np.abs(c - np.cos(x.astype(np.float64))).max()),
                         "replaces": "__sinf/__cosf (cudaSiftD.cu:331-332), CUDA documents 2^-21.41 = 3.6e-7 absolute on [-pi, pi]"}
    path = os.path.join(ROOT, "profiles", "r03_det_accuracy.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
