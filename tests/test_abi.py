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


def test_descriptor_sizes_match_header_layout(tmp_path):
    """The ctypes mirrors have exactly the sizes (and the offsets of their last fields) that a C compiler gives the
    structs of include/dsl_hip.h: the header is compiled with gcc and asked."""
    import subprocess
    from dsl_amd import _lib as L
    pairs = [('dsl_conv_desc', L.ConvDesc, 'gn_stats'), ('dsl_wgrad_desc', L.WgradDesc, 'pixtab_bytes'),
             ('dsl_gn_desc', L.GnDesc, 'y8_amax'), ('dsl_fp8_prep_item', L.Fp8PrepItem, 'cout'), ('dsl_fcos_desc', L.FcosDesc, 'logvec'),
             ('dsl_det_desc', L.DetDesc, 'workspace_bytes'), ('dsl_pack_item', L.PackItem, 'tiles_co'), ('dsl_op', L.Op, 'l'),
             ('dsl_rec_sum_item', L.RecSumItem, 'nrec'), ('dsl_bn_post_item', L.BnPostItem, 'row_start')]
    src = tmp_path / 'sizes.c'
    body = ''.join(f'  printf("%zu %zu\\n", sizeof({c}), offsetof({c}, {f}));\n' for c, _, f in pairs)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dsl_hip.h"\nint main(void) {\n' + body + '  return 0;\n}\n')
    exe = tmp_path / 'sizes'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split()
    for i, (c, t, f) in enumerate(pairs):
        assert ctypes.sizeof(t) == int(out[2 * i]), (c, ctypes.sizeof(t), out[2 * i])
        assert getattr(t, f).offset == int(out[2 * i + 1]), (c, f)


def test_errors_are_reported_not_swallowed():
    from dsl_amd import _lib as L
    d = L.ConvDesc()          # all zero: invalid
    rc = L.lib.dsl_conv2d(ctypes.byref(d), None)
    assert rc != 0 and b'nseg' in L.lib.dsl_last_error()
