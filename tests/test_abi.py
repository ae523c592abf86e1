"""The C-ABI library loads and exports every symbol include/dsl_hip.h declares (no compute, CPU only)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    txt = open(os.path.join(ROOT, 'include', 'dsl_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(dsl_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_exported():
    from dsl_amd import _lib
    names = declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(_lib.lib, n)]
    assert not missing, missing
    assert not _lib.MISSING
    assert _lib.lib.dsl_version() >= 100
    assert isinstance(_lib.lib.dsl_last_error(), bytes)


def test_descriptor_sizes_match_header_layout():
    """ctypes mirrors must have the C struct sizes (natural alignment, LP64)."""
    from dsl_amd import _lib as L
    assert ctypes.sizeof(L.ConvDesc) == 2 * 4 + 8 * 5 * 4 + 13 * 4 + 4 + 9 * 8     # 4 bytes padding before the pointers
    assert ctypes.sizeof(L.Op) == 4 + 7 * 4 + 8 + 4 * 8 + 2 * 8
    assert ctypes.sizeof(L.GnDesc) % 8 == 0 and ctypes.sizeof(L.FcosDesc) % 8 == 0


def test_errors_are_reported_not_swallowed():
    from dsl_amd import _lib as L
    d = L.ConvDesc()          # all zero: invalid
    rc = L.lib.dsl_conv2d(ctypes.byref(d), None)
    assert rc != 0 and b'nseg' in L.lib.dsl_last_error()
