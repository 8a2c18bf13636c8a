"""The C ABI (include/aewavenet.h -> ae_wavenet_amd/lib/libaewavenet_hip.so) without a GPU: the library loads, exports
every function the header declares, reports the header's ABI version, and its struct sizes agree with the ctypes
mirrors of ae_wavenet_amd/_lib.py.  No compute entry point is called."""
import ctypes as C
import os
import re

import pytest

from ae_wavenet_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "aewavenet.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)                 # comments out
    body = src[src.index('extern "C"'):]
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|char)\s*\*?\s*(aew_[a-z0-9_]+)\s*\(", body, flags=re.M)
    return sorted(set(names))


def test_library_exports_every_declared_function():
    if not os.path.exists(L.LIB_PATH):
        pytest.fail(f"{L.LIB_PATH} missing: run __graft_entry__.build() first")
    lib = C.CDLL(L.LIB_PATH)                                         # plain dlopen: no device needed
    names = declared_functions()
    assert len(names) >= 20 and "aew_run_plan" in names and "aew_sampler_run" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the binding's own list is the header's list
    assert sorted(L.EXPORTS) == names, sorted(set(names) ^ set(L.EXPORTS))


def test_abi_version_and_struct_sizes():
    lib = L.load()                                                   # raises on version / size drift
    version = int(re.search(r"#define\s+AEW_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.aew_abi_version() == version
    for which, cls in ((0, L.Op), (1, L.GemmNT), (2, L.GemmTN), (3, L.Seg), (4, L.View), (5, L.CopyRec), (6, L.Actor),
                       (7, L.Sampler), (8, L.Tuning)):
        assert lib.aew_sizeof(which) == C.sizeof(cls), cls.__name__
    assert lib.aew_sizeof(99) == -1
    assert lib.aew_strerror(-1) and lib.aew_strerror(0)


def test_tuning_record_defaults_get_set_and_clamping():
    """aew_tuning_t (ABI 17): the aew_set_* switches edit the process-wide record, aew_tuning_get reads it back,
    aew_tuning_set replaces it (clamped like the setters), aew_tuning_default never changes.  Host logic only."""
    lib = L.load()
    d, cur = L.Tuning(), L.Tuning()
    assert lib.aew_tuning_default(C.byref(d)) == 0 and lib.aew_tuning_get(C.byref(cur)) == 0
    assert (d.nt_wave_rows, d.nt_window, d.nt_rows192, d.lanes, d.fn_enable, d.tn_target_blocks) == (64, 64, 1, 0, 1, 512)
    assert bytes(d) == bytes(cur), "the process-wide record starts at the defaults"
    try:
        lib.aew_set_nt_window(16)
        lib.aew_set_lanes(7)                                     # clamped to 2
        lib.aew_tuning_get(C.byref(cur))
        assert (cur.nt_window, cur.lanes) == (16, 2)
        t = L.default_tuning(nt_wave_rows=77, nt_mem128=9, tn_target_blocks=0)     # nonsense values: clamped / reset
        assert lib.aew_tuning_set(C.byref(t)) == 0
        lib.aew_tuning_get(C.byref(cur))
        assert (cur.nt_wave_rows, cur.nt_mem128, cur.tn_target_blocks, cur.nt_window) == (64, 2, 1, 64)
        lib.aew_tuning_default(C.byref(cur))
        assert bytes(cur) == bytes(d)
        with pytest.raises(AttributeError):
            L.default_tuning(no_such_field=1)
    finally:
        lib.aew_tuning_set(C.byref(d))
    lib.aew_tuning_get(C.byref(cur))
    assert bytes(cur) == bytes(d)


def test_missing_library_fails_loudly(monkeypatch):
    """No CPU fallback: without the .so the product path raises instead of computing something else."""
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", L.LIB_PATH + ".absent")
    with pytest.raises(L.AewError):
        L.load()


def test_nt_kernel_query_follows_the_dispatch_rules():
    """aew_nt_kernel (host-side logic only): which kernel a GEMM_NT descriptor runs on.  bench.py groups its roofline
    by it, so it has to follow launch_gemm_nt: large bf16 launches -> the tiled kernel, launches of few tiles -> the
    64-row shape, impl = 2 on a covered shape -> the full-N kernel, fp32 -> the exact-chain kernel, impl = 1 -> check."""
    from ae_wavenet_amd.plan import Mat, Workspace, make_nt
    lib = L.load()
    ws = Workspace("cpu")
    x = Mat.new(ws, "x", 8, 6100, 384, L.BF16)
    W = Mat.new(ws, "W", 1, 512, 768, L.BF16)
    y = Mat.new(ws, "y", 8, 6100, 512, L.BF16)
    segs = [x.seg(384), x.seg(384, row_off=16)]
    q = lambda g: lib.aew_nt_kernel(C.byref(g))
    # two taps of one tensor, 16 rows apart: the one-window kernel; 128 rows apart (or with it switched off): the
    # two-segment kernel
    far = [x.seg(384), x.seg(384, row_off=128)]
    assert q(make_nt(L.BF16, 6000, 512, 512, 8, segs, W.ptr, out0=y.view())) == 6
    assert q(make_nt(L.BF16, 5900, 512, 512, 8, far, W.ptr, out0=y.view())) == 0
    lib.aew_set_nt_window(8)
    try:
        assert q(make_nt(L.BF16, 6000, 512, 512, 8, segs, W.ptr, out0=y.view())) == 0
    finally:
        lib.aew_set_nt_window(64)
    assert q(make_nt(L.BF16, 6000, 512, 512, 8, segs, W.ptr, out0=y.view(), impl=2)) == 2
    assert q(make_nt(L.BF16, 6000, 512, 512, 8, segs, W.ptr, out0=y.view(), impl=1)) == 4
    assert q(make_nt(L.BF16, 70, 512, 512, 8, segs, W.ptr, out0=y.view())) == 1           # 8 x 4 tiles of 256 rows
    lib.aew_set_nt_small_tiles(0)
    try:
        assert q(make_nt(L.BF16, 70, 512, 512, 8, far, W.ptr, out0=y.view())) == 0
    finally:
        lib.aew_set_nt_small_tiles(128)
    lib.aew_set_fn(0)
    try:
        assert q(make_nt(L.BF16, 5900, 512, 512, 8, far, W.ptr, out0=y.view(), impl=2)) == 0
    finally:
        lib.aew_set_fn(1)
    xf = Mat.new(ws, "xf", 8, 80, 768, L.F32)
    Wf = Mat.new(ws, "Wf", 1, 768, 768, L.F32)
    yf = Mat.new(ws, "yf", 8, 80, 768, L.F32)
    assert q(make_nt(L.F32, 70, 768, 768, 8, [xf.seg(768)], Wf.ptr, out0=yf.view())) == 3
    assert lib.aew_nt_kernel(None) == -1
