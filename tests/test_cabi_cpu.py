"""CPU tests of the boundary (no GPU, no compute calls): the C-ABI library loads and exports every symbol
include/misift.h declares; the C++ shim exports the reference's mangled C++ API; struct layouts match."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def declared_functions():
    src = open(os.path.join(INC, "misift.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(misift_[a-z0-9_]+)\s*\(", src)))


def test_cabi_exports_every_declared_symbol():
    from cudasift_amd import capi
    L = capi.lib()                       # raises if libmisift.so is missing: there is no fallback
    names = declared_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), "libmisift.so does not export %s" % n
    # the ctypes table covers the whole header (so tests/bench bind by the declared signatures)
    assert set(names) == set(capi.SIGNATURES.keys()), set(names) ^ set(capi.SIGNATURES.keys())


def test_cabi_no_device_paths_fail_cleanly():
    """On a box without a GPU nothing falls back to the CPU: context creation reports MISIFT_ENODEV."""
    from cudasift_amd import capi
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.MisiftError):
        capi.Context(0)
    assert b"no HIP device" in capi.lib().misift_last_error()


def test_host_only_entry_points():
    from cudasift_amd import capi
    # scratch sizing follows cudaSiftH.cu:39-57 (numOctaves+1 levels), rounded to 4096 floats
    w, h, p = 1920, 1080, 1920
    size, tmp = h * p, 8 * h * p
    ww, hh = w, h
    for _ in range(5):
        ww //= 2
        hh //= 2
        pp = (ww + 127) // 128 * 128
        size += hh * pp
        tmp += 8 * hh * pp
    assert capi.scratch_floats(1920, 1080, 5) == (size + tmp + 4095) // 4096 * 4096
    from oracle import pyoracle as orc
    assert np.array_equal(capi.laplace_taps(5), orc.laplace_taps(5))     # same host formula, same libm
    o = capi.Options()
    capi.lib().misift_default_options_sized(C.byref(o), C.sizeof(o))
    assert (o.texfrac_bits, o.fix_numpts, o.match_full, o.match_exact_top2, o.reference_cap) == (8, 0, 0, 0, 0)
    # the struct grows at its end and carries no size: the _sized entry points never touch bytes past the caller's struct,
    # and the plain symbols (binaries built against the r04 header) stop after that header's seven fields
    o = capi.Options()
    o.reference_cap = 77
    capi.lib().misift_default_options(C.byref(o))
    assert o.texfrac_bits == 8 and o.reference_cap == 77
    o.texfrac_bits = -1
    capi.lib().misift_default_options_sized(C.byref(o), 4 * 4)
    assert o.texfrac_bits == 8 and o.reference_cap == 77


def test_point_record_layout():
    from cudasift_amd import capi
    d = capi.POINT_DTYPE
    assert d.itemsize == 576
    assert d.fields["score"][1] == 24 and d.fields["match"][1] == 32 and d.fields["match_xpos"][1] == 36
    assert d.fields["subsampling"][1] == 48 and d.fields["data"][1] == 64


def test_cpp_shim_exports_reference_api():
    lib = os.path.join(ROOT, "cudasift_amd", "libcudasift.so")
    assert os.path.exists(lib), "run `make`"
    syms = subprocess.check_output("nm -D --defined-only %s | c++filt" % lib, shell=True, text=True)
    for want in ["InitCuda(int)", "AllocSiftTempMemory(int, int, int, bool)", "FreeSiftTempMemory(float*)",
                 "ExtractSift(SiftData&, CudaImage&, int, double, float, float, bool, float*)",
                 "InitSiftData(SiftData&, int, bool, bool)", "FreeSiftData(SiftData&)", "PrintSiftData(SiftData&)",
                 "MatchSiftData(SiftData&, SiftData&)",
                 "FindHomography(SiftData&, float*, int*, int, float, float, float)",
                 "CudaImage::CudaImage()", "CudaImage::~CudaImage()",
                 "CudaImage::Allocate(int, int, int, bool, float*, float*)", "CudaImage::Download()",
                 "CudaImage::Readback()", "CudaImage::InitTexture()", "CudaImage::CopyToTexture(CudaImage&, bool)",
                 "iDivUp(int, int)", "iDivDown(int, int)", "iAlignUp(int, int)", "iAlignDown(int, int)"]:
        assert want in syms, want


def test_headers_are_plain_cpp_and_layout_matches(tmp_path):
    src = tmp_path / "layout.cpp"
    src.write_text('#include <cstdio>\n#include <cstddef>\n#include "cudaSift.h"\n'
                   'int main(){ CudaImage img; SiftData d; (void)d;\n'
                   ' printf("%zu %zu %zu %zu %zu %zu %d\\n", sizeof(SiftPoint), offsetof(SiftPoint,score),'
                   ' offsetof(SiftPoint,match), offsetof(SiftPoint,subsampling), offsetof(SiftPoint,data),'
                   ' sizeof(SiftData), img.pitch); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["g++", "-std=c++17", "-I", INC, str(src), "-o", str(exe), "-L",
                           os.path.join(ROOT, "cudasift_amd"), "-lcudasift", "-lmisift",
                           "-Wl,-rpath," + os.path.join(ROOT, "cudasift_amd")])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert out == ["576", "24", "32", "48", "64", "24", "0"]


def test_managedmem_flavour_builds_and_exports(tmp_path):
    """The reference's alternate ABI (cudaSift.h:27-32, -DMANAGEDMEM: SiftData = {numPts, maxPts, m_data}): the header
    compiles in that mode with the same record layout, libcudasift_managed.so exports the same API, and the
    reference's unmodified mainSift.cpp + geomFuncs.cpp link against it."""
    lib = os.path.join(ROOT, "cudasift_amd", "libcudasift_managed.so")
    assert os.path.exists(lib), "run `make`"
    syms = subprocess.check_output("nm -D --defined-only %s | c++filt" % lib, shell=True, text=True)
    for want in ["ExtractSift(SiftData&, CudaImage&, int, double, float, float, bool, float*)",
                 "InitSiftData(SiftData&, int, bool, bool)", "MatchSiftData(SiftData&, SiftData&)",
                 "FindHomography(SiftData&, float*, int*, int, float, float, float)",
                 "ImproveHomographyGPU(SiftData&, float*, int, float, float, float)"]:
        assert want in syms, want
    src = tmp_path / "layout.cpp"
    src.write_text('#include <cstdio>\n#include <cstddef>\n#include "cudaSift.h"\n'
                   'int main(){ SiftData d; d.m_data = 0; printf("%zu %zu %zu\\n", sizeof(SiftPoint), sizeof(SiftData),'
                   ' offsetof(SiftData, m_data)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["g++", "-std=c++17", "-DMANAGEDMEM", "-I", INC, str(src), "-o", str(exe)])
    assert subprocess.check_output([str(exe)], text=True).split() == ["576", "16", "8"]
    ref = os.environ.get("REF", "/root/reference")
    if os.path.exists(os.path.join(ref, "mainSift.cpp")):
        exe2 = tmp_path / "dropin_managed"
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-DMANAGEDMEM", "-I", INC, "-I",
                               os.path.join(ROOT, "cudasift_amd", "compat"), os.path.join(ref, "mainSift.cpp"),
                               os.path.join(ref, "geomFuncs.cpp"), "-o", str(exe2), "-L",
                               os.path.join(ROOT, "cudasift_amd"), "-lcudasift_managed", "-lmisift",
                               "-Wl,-rpath," + os.path.join(ROOT, "cudasift_amd")])
        assert os.path.exists(exe2)


def test_reference_main_compiles_unchanged(tmp_path):
    ref = os.environ.get("REF", "/root/reference")
    if not os.path.exists(os.path.join(ref, "mainSift.cpp")):
        pytest.skip("reference tree absent")
    exe = tmp_path / "dropin"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", INC, "-I", os.path.join(ROOT, "cudasift_amd", "compat"),
                           os.path.join(ref, "mainSift.cpp"), os.path.join(ref, "geomFuncs.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "cudasift_amd"), "-lcudasift", "-lmisift",
                           "-Wl,-rpath," + os.path.join(ROOT, "cudasift_amd")])
    assert os.path.exists(exe)


def test_header_is_plain_c99(tmp_path):
    """include/misift.h is the FFI surface: it must compile as strict C99 (no C++ constructs, no HIP types)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "misift.h"\nint main(void) { misift_options o; (void)o; return 0; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_frame_shares_of_the_balanced_keypoint_launches():
    """misift_test_frame_shares (host-only; the formula frame_shares_kernel evaluates on the device, MISIFT_BALANCE=1):
    every frame keeps at least one workgroup (the first sub-block of a frame publishes its counters), the shares never add
    up to more than the launch has, they follow the keypoint counts, and equal counts get equal shares."""
    from cudasift_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(11)

    def shares(nblocks, pts):
        pts = np.ascontiguousarray(pts, np.uint32)
        out = np.zeros(len(pts), np.int32)
        assert L.misift_test_frame_shares(nblocks, len(pts), pts.ctypes.data, out.ctypes.data) == 0
        return out

    cases = [(2048, [2000] * 64), (2048, [0] * 64), (512, [30000] + [0] * 63), (1280, [16709] + [250] * 63),
             (8, [5]), (64, [0, 1, 2, 3, 4, 5, 6, 7]), (8 * 300, list(rng.integers(0, 40000, 300))),
             (4096, [163840] * 512), (1024, [1] + [0] * 127)]
    cases += [(int(n * rng.integers(8, 40)), list(rng.integers(0, 5 * 32768, n) * (rng.random(n) < 0.7)))
              for n in rng.integers(5, 600, 100)]
    for nblocks, pts in cases:
        pts = np.array(pts, np.uint64)
        s = shares(nblocks, pts)
        assert s.min() >= 1 and s.sum() <= nblocks, (nblocks, s.sum())
        total = int(pts.sum())
        if total:
            spare = nblocks - len(pts)
            assert np.array_equal(s, 1 + (spare * pts) // total)          # the documented formula, exactly
            assert s.sum() > nblocks - len(pts)                           # at most one workgroup per frame left unused
            order = np.argsort(pts, kind="stable")
            assert np.all(np.diff(s[order]) >= 0)                         # more keypoints, never fewer workgroups
        else:
            assert np.all(s == 1)
    assert L.misift_test_frame_shares(3, 4, np.zeros(4, np.uint32).ctypes.data, np.zeros(4, np.int32).ctypes.data) != 0
    assert L.misift_test_frame_shares(8, 0, None, None) != 0


def test_matcher_chunk_plan_properties():
    """misift_test_match_plan (host-only): the column-chunk plan of the matcher covers every super-tile exactly once, has
    no empty chunk, survives degenerate shapes (n2 < 32: no column takes part; the division by zero of an earlier
    attempt), and for big problems stays within 5 % of the ideal number of CU rounds x tiles."""
    import ctypes as C
    from cudasift_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(5)
    shapes = [(100000, 100000), (12500, 100000), (25000, 100000), (16384, 16384), (2000, 2000), (1, 33), (5, 20), (0, 100),
              (333, 64), (70, 31), (1, 1), (128, 64), (129, 65)]
    shapes += [(int(rng.integers(1, 200000)), int(rng.integers(1, 200000))) for _ in range(200)]
    for n1, n2 in shapes:
        for cus in (256, 64, 304):
            a, b, c = C.c_int(), C.c_int(), C.c_int()
            assert L.misift_test_match_plan(cus, n1, n2, C.byref(a), C.byref(b), C.byref(c)) == 0
            nch, tpc, nt = a.value, b.value, c.value
            assert nt == (32 * (n2 // 32) + 63) // 64
            assert nch >= 1 and tpc >= 1
            if nt > 0:
                assert nch * tpc >= nt and (nch - 1) * tpc < nt, (n1, n2, nch, tpc, nt)
                nrb = (n1 + 127) // 128
                if nrb * nt >= 40 * cus and nrb > 0:          # enough work for the quantisation to be a detail
                    rounds = -(-nrb * nch // cus)
                    assert rounds * tpc <= 1.05 * nrb * nt / cus + tpc, (n1, n2, cus, nch, tpc)


def test_every_environment_knob_of_the_library_is_documented():
    """Each getenv("MISIFT_...") in csrc/ is named in README.md, INTEGRATION.md or include/misift.h."""
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    knobs = set()
    for src in (root / "cudasift_amd" / "csrc").iterdir():
        if src.suffix in (".hip", ".cpp", ".hpp"):
            text = src.read_text()
            knobs |= set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', text))
            knobs |= set(re.findall(r'match_plan_param\("([A-Z_0-9]+)"', text))
            knobs |= set(re.findall(r'\{"[a-z_]+", "(MISIFT_[A-Z_0-9]+)"\}', text))       # the knob table (MISIFT_TUNABLES=1)
    docs = "".join((root / f).read_text() for f in ("README.md", "INTEGRATION.md", "include/misift.h"))
    missing = sorted(k for k in knobs if k not in docs)
    assert len(knobs) > 20 and not missing, missing
    # the knob table and the test entry point agree, and the library exports it without a GPU
    from cudasift_amd import capi
    names = capi.knob_names()
    assert len(names) >= 25 and all(v in knobs for v in names.values()) and "MISIFT_TUNABLES" in knobs
