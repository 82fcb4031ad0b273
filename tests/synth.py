"""Synthetic inputs for tests and bench: textured frames and match.cu-style descriptors."""
import numpy as np


SYNTH_AMP = 26.0      # tuned once so the oracle finds ~2000 keypoints per 1920x1080 frame (thresh 3.0), then frozen
SYNTH_BANDS = 6


def synth_frame(f, width=1920, height=1080, amp=SYNTH_AMP):
    """Band-limited noise frame, fp32 in [0,255], fully textured (no flat patches).

    frame f = clamp(128 + amp * S / std(S), 0, 255), S = sum_{j=0..5} (G_{sigma=2^j} * W_{f,j}) / std_j
    (equal energy per octave band, like natural 1/f images), W iid N(0,1) from PCG64 seeded
    0x51F7 + f.  The Gaussian blurs are applied in the Fourier domain (periodic boundary)."""
    rng = np.random.Generator(np.random.PCG64(0x51F7 + int(f)))
    fy = np.fft.fftfreq(height)[:, None]
    fx = np.fft.rfftfreq(width)[None, :]
    r2 = fx * fx + fy * fy
    acc = np.zeros((height, width), np.float64)
    for j in range(SYNTH_BANDS):
        w = rng.standard_normal((height, width))
        sigma = 2.0 ** j
        g = np.exp(-2.0 * (np.pi ** 2) * (sigma ** 2) * r2)
        b = np.fft.irfft2(np.fft.rfft2(w) * g, s=(height, width))
        acc += b / b.std()
    img = 128.0 + amp * acc / acc.std()
    return np.clip(img, 0.0, 255.0).astype(np.float32)


def synth_descriptors(n, seed=12345, l2=False):
    """match.cu:945-957 recipe: uniform [0,1) vectors scaled by sqrt(128)/sum (or L2-normalised)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    d = rng.random((n, 128), dtype=np.float32)
    if l2:
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    else:
        d *= (np.float32(np.sqrt(128.0)) / d.sum(axis=1, keepdims=True, dtype=np.float32))
    return d.astype(np.float32)


def descriptors_to_points(d, dtype):
    """Embed [n,128] descriptors into SiftPoint records (AoS, 576-byte stride)."""
    pts = np.zeros(d.shape[0], dtype)
    pts["data"] = d
    pts["xpos"] = np.arange(d.shape[0], dtype=np.float32) % 1920
    pts["ypos"] = np.arange(d.shape[0], dtype=np.float32) // 1920
    pts["match"] = -2
    return pts


def synth_matches(n, inlier_frac=0.6, seed=0, width=1920, height=1080, noise=0.7, dtype=None):
    """A SiftPoint array whose stored matches follow a known homography for `inlier_frac` of the
    points (plus `noise` px jitter) and are random for the rest; scores/ambiguities are mixed so
    that FindHomography's validity filter (matching.cu:1033-1037) selects an ordered subset.
    Returns (points, H_true 3x3, inlier mask)."""
    from cudasift_amd.capi import POINT_DTYPE
    rng = np.random.default_rng(seed)
    pts = np.zeros(n, dtype or POINT_DTYPE)
    x = rng.uniform(0, width, n).astype(np.float32)
    y = rng.uniform(0, height, n).astype(np.float32)
    H = np.array([[0.98, 0.03, 14.0], [-0.025, 1.01, -9.0], [1.5e-5, -1.0e-5, 1.0]], np.float64)
    den = H[2, 0] * x + H[2, 1] * y + 1.0
    mx = (H[0, 0] * x + H[0, 1] * y + H[0, 2]) / den
    my = (H[1, 0] * x + H[1, 1] * y + H[1, 2]) / den
    inl = rng.uniform(size=n) < inlier_frac
    mx = np.where(inl, mx + rng.normal(0, noise, n), rng.uniform(0, width, n))
    my = np.where(inl, my + rng.normal(0, noise, n), rng.uniform(0, height, n))
    pts["xpos"], pts["ypos"] = x, y
    pts["match_xpos"], pts["match_ypos"] = mx.astype(np.float32), my.astype(np.float32)
    pts["score"] = np.where(inl, rng.uniform(0.86, 1.0, n), rng.uniform(0.5, 1.0, n)).astype(np.float32)
    pts["ambiguity"] = np.where(inl, rng.uniform(0.3, 0.99, n), rng.uniform(0.6, 1.0, n)).astype(np.float32)
    pts["match"] = rng.integers(0, max(n, 1), n)
    return pts, H.astype(np.float32), inl


def dense_extrema_dog(w=128, h=64, height=10.0):
    """A stack of 7 DoG planes [7, h, w] with MORE scale-space maxima per 30x8 block than the reference keeps: plane 3
    holds isolated peaks on a (2, 3)-pixel lattice (1/6 of the pixels, ~40 per block), everything else is zero, so every
    lattice point is a strict 26-neighbour maximum that passes the edge test (tr^2 = 4 det / ... < 10 det).
    Returns (dog, lattice ys, lattice xs)."""
    dog = np.zeros((7, h, w), np.float32)
    ys, xs = np.meshgrid(np.arange(2, h - 2, 3), np.arange(2, w - 2, 2), indexing="ij")
    dog[3, ys, xs] = height
    return dog, ys.ravel(), xs.ravel()
