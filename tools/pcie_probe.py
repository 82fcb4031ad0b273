#!/usr/bin/env python3
"""Developer aid: raw PCIe copy rates (pinned host memory), each direction alone and both together."""
import time, torch
dev = torch.device("cuda:0")
n = 128 << 20
h1 = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device=dev); d2 = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, reps=8):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return reps * n / (time.perf_counter() - t0) / 1e9
def h2d():
    with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
def both():
    h2d(); d2h()
print("H2D %.1f GB/s  D2H %.1f GB/s  both (each direction) %.1f GB/s" % (t(h2d), t(d2h), t(both)))
