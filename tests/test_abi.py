"""The C-ABI library builds, loads and exports every symbol include/specmi.h declares.
No compute is run here (no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from spec_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'specmi.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(specmi_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_all_exported(lib):
    from spec_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in specmi.h but not exported'
        assert name in _lib.PROTOTYPES, f'{name} has no ctypes prototype'
    assert sorted(_lib.PROTOTYPES) == declared


def test_version_and_no_gpu_error_path(lib):
    import torch
    assert b'gfx950' in lib.specmi_version()
    if torch.cuda.is_available():
        pytest.skip('GPU present: error path for a missing device not reachable')
    h = C.c_void_p()
    rc = lib.specmi_create(C.byref(h), 0, 1)
    assert rc != 0 and not h.value
    assert b'HIP' in lib.specmi_last_error(None) or b'device' in lib.specmi_last_error(None)


def test_struct_layouts_match_header():
    from spec_amd import _lib
    assert C.sizeof(_lib.HmrOutputs) == 8 * C.sizeof(C.c_void_p)
    assert C.sizeof(_lib.ProfEntry) == 48 + 48 + 3 * 8 + 8  # int + padding


def test_product_path_has_no_cpu_fallback():
    import torch
    from spec_amd import assets
    from spec_amd.modules import CameraRegressorNetwork
    m = CameraRegressorNetwork()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 224, 224))
    # the product package never imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'spec_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


# ---- the option surface (round 6: frozen stable list, gated experimental list) ---------------------------------------------
def _option_table(lib):
    out, i = {}, 0
    while True:
        name, d, st = C.c_char_p(), C.c_int(), C.c_int()
        if lib.specmi_option_info(i, C.byref(name), C.byref(d), C.byref(st)) != 0:
            return out
        out[name.value.decode()] = (d.value, bool(st.value))
        i += 1


STABLE = {'backbone': 50, 'num_fc_layers': 1, 'num_fc_channels': 1024, 'use_cam': 0, 'use_cam_feats': 0, 'img_res': 224, 'hrnet_use_conv': 1,
          'estimate_var': 0, 'uncertainty_activation': 0,
          'plan': 0, 'winograd': 1, 'fuse_downsample': 1, 'head_collapse': 1, 'output_ld': 0, 'angle_ld': 0, 'experimental': 0}


def test_option_table_matches_header_and_call_sites(lib):
    """The library's option table is the single list: its stable part is exactly the frozen set with the documented defaults, every
    name (and nothing else) is documented in include/specmi.h under the right heading, and the default of every ``opt_i(h, "name", N)``
    call site in the sources equals the table's."""
    table = _option_table(lib)
    assert {k: v[0] for k, v in table.items() if v[1]} == STABLE
    hdr = open(os.path.join(ROOT, 'include', 'specmi.h')).read()
    a, b = hdr.index(' * STABLE ('), hdr.index(' * EXPERIMENTAL (')
    stable_doc, exp_doc = hdr[a:b], hdr[b:hdr.index('int specmi_set_option_i32')]
    quoted = lambda text: set(re.findall(r'"([a-z][a-z_0-9]*)"', text))
    for name, (_, stable) in table.items():
        assert name in quoted(stable_doc if stable else exp_doc), f'option {name} is not documented under its heading'
    assert quoted(stable_doc) - {'focal_length'} <= {k for k, v in table.items() if v[1]}, quoted(stable_doc) - set(table)
    assert quoted(exp_doc) - {'experimental'} <= {k for k, v in table.items() if not v[1]}, quoted(exp_doc) - set(table)
    csrc = os.path.join(ROOT, 'spec_amd', 'csrc')
    seen = set()
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(('.hip', '.h', '.inc')):
            continue
        for name, dflt in re.findall(r'opt_i\(h[ab]?, "([a-z_0-9]+)", ([^)]*)\)', open(os.path.join(csrc, f)).read()):
            assert name in table, f'{f}: opt_i reads an option the table does not list: {name}'
            seen.add(name)
            if re.fullmatch(r'-?\d+', dflt.strip()):
                assert int(dflt) == table[name][0], f'{f}: call-site default of {name} is {dflt}, the table says {table[name][0]}'
    assert seen == set(table) - {'experimental'}, set(table) ^ seen
