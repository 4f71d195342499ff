"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/ctl_hip.h declares, the
struct layouts match the ctypes/numpy mirrors, and argument validation fails loudly (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from cooperative_training_and_latent_space_data_augmentation_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_reports_version():
    header = open(os.path.join(ROOT, "include", "ctl_hip.h")).read()
    declared = int(re.search(r"#define\s+CTL_ABI_VERSION\s+(\d+)", header).group(1))
    assert _ffi.lib.ctl_version() == declared == _ffi.ABI_VERSION          # a binding written against another layout is refused at load
    assert _ffi.lib.ctl_red_blocks() == _ffi.RED_BLOCKS


def test_stale_binding_is_refused(monkeypatch):
    """a consumer built against an older header must get an error at load, not mis-laid-out structs."""
    fresh = _ffi._Lib()
    monkeypatch.setattr(_ffi, "ABI_VERSION", _ffi.ABI_VERSION - 1)
    with pytest.raises(_ffi.CtlError, match="ABI mismatch"):
        fresh.load()


def test_shipped_library_reads_no_environment():
    """tuning hooks exist only in -DCTL_TUNING builds; the default library neither imports getenv nor carries the
    names of the hooks."""
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms, [l for l in syms.splitlines() if "getenv" in l]
    blob = open(_ffi.LIB_PATH, "rb").read()
    for name in (b"CTL_FORCE_CFG", b"CTL_PERSIST", b"CTL_WGRAD_SLOTS", b"CTL_SIDE_STREAM", b"CTL_FUSE_FINALIZE", b"CTL_FUSE_CONSUMER",
                 b"CTL_MASK_SPLIT", b"CTL_PROF_TIMELINE", b"CTL16_WGRAD_SPLITS"):
        assert name not in blob, name
    pkg = os.path.join(ROOT, "cooperative_training_and_latent_space_data_augmentation_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "os.environ" not in open(os.path.join(pkg, fn)).read(), fn


def test_every_declared_symbol_is_exported():
    header = open(os.path.join(ROOT, "include", "ctl_hip.h")).read()
    declared = set(re.findall(r"\b(ctl_[a-z0-9_]+)\s*\(", header))
    declared -= {"ctl_stream"}
    raw = ctypes.CDLL(_ffi.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(raw, s)]
    assert not missing, missing
    assert declared == set(_ffi.EXPORTED), declared ^ set(_ffi.EXPORTED)


def test_struct_layouts_match():
    assert _ffi.lib.ctl_sizeof_op() == _ffi.OP_DTYPE.itemsize == 328
    assert _ffi.lib.ctl_sizeof_conv() == _ffi.CONV_DTYPE.itemsize == 96


def test_invalid_arguments_fail_loudly():
    d = _ffi.conv_desc(n=1, hin=8, win=8, cin=16, hout=8, wout=8, cout=16, ks=3)
    rc = _ffi.lib.ctl_conv_forward(_ffi.desc_ptr(d), None, None, None, None, None, None, None, None, None, None, None)
    assert rc == -1 and b"null" in _ffi.lib.ctl_last_error()
    with pytest.raises(_ffi.CtlError):
        _ffi.check(rc, "ctl_conv_forward")
    bad = _ffi.conv_desc(n=1, hin=8, win=8, cin=24, hout=8, wout=8, cout=16, ks=3)   # cin >= 16 must be a multiple of 16
    assert _ffi.lib.ctl_conv_stats_blocks(_ffi.desc_ptr(bad)) == -1
    bad2 = _ffi.conv_desc(n=1, hin=8, win=8, cin=16, hout=8, wout=8, cout=16, ks=3, stride=2, in_mode=_ffi.IN_UP2)
    assert _ffi.lib.ctl_conv_stats_blocks(_ffi.desc_ptr(bad2)) == -1


def test_prologue_coefficient_table_limit_is_enforced():
    """The BatchNorm prologue coefficients live in a 256-entry LDS table per block (groups * cin): more is refused, not truncated."""
    import ctypes
    dummy = ctypes.cast((ctypes.c_float * 4)(), ctypes.c_void_p)          # never dereferenced: the check fails before any launch
    d = _ffi.conv_desc(n=4, hin=8, win=8, cin=128, hout=8, wout=8, cout=16, ks=3, groups=4, pro_affine=1, pro_slope=0.2)
    rc = _ffi.lib.ctl_conv_forward(_ffi.desc_ptr(d), dummy, dummy, None, dummy, dummy, None, None, None, dummy, None, None)
    assert rc == -1 and b"prologue coefficients" in _ffi.lib.ctl_last_error()
    rc = _ffi.lib.ctl_conv_wgrad(_ffi.desc_ptr(d), dummy, dummy, dummy, dummy, dummy, dummy, None)
    assert rc == -1 and b"prologue coefficients" in _ffi.lib.ctl_last_error()


def test_host_side_sizing_helpers():
    assert _ffi.lib.ctl_conv_wpack_floats(16, 16, 3) == 9 * 256
    assert _ffi.lib.ctl_conv_wpack_floats(4, 16, 3) == 9 * 256          # cin padded to one 16-chunk
    assert _ffi.lib.ctl_conv_wpack_floats(128, 64, 1) == 4 * 8 * 256
    d = _ffi.conv_desc(n=16, hin=256, win=256, cin=16, hout=256, wout=256, cout=16, ks=3)
    # persistent grid = 256 CUs x resident blocks per CU of the chosen kernel (asked from the runtime; 2 assumed without a GPU)
    blocks = _ffi.lib.ctl_conv_stats_blocks(_ffi.desc_ptr(d))
    assert blocks in (512, 768, 1024)
    assert _ffi.lib.ctl_conv_stats_floats(_ffi.desc_ptr(d)) == blocks * 2 * 16
    assert _ffi.lib.ctl_wgrad_splits(_ffi.desc_ptr(d)) == 256          # one block per CU (round 2: the weight gradients co-run with the other chain)
    assert _ffi.lib.ctl_wgrad_partial_floats(_ffi.desc_ptr(d)) == 256 * 9 * 16 * 16
    assert _ffi.lib.ctl_latent_score_ws_floats(0, 16, 256, 128) == 16 * 4 * 128


def test_extension_binds_to_torchs_hip_runtime():
    """Loading the extension imports torch first (one HIP runtime per process: the wheel's libamdhip64.so, not /opt/rocm's): checked in
    a fresh interpreter that touches the extension before anything else -- build() followed by smoke() in one process."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from cooperative_training_and_latent_space_data_augmentation_amd import _ffi\n"
            "assert 'torch' not in sys.modules\n"
            "assert _ffi.lib.ctl_version() >= 1\n"
            "assert 'torch' in sys.modules\n"
            "maps = open('/proc/self/maps').read()\n"
            "hips = sorted({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l})\n"
            "assert len(hips) == 1 and '/torch/lib/' in hips[0], hips\n"
            "print('OK', hips[0])\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
