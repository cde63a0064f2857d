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
