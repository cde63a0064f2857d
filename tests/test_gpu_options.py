"""The frozen option surface of include/specmi.h on a live handle: effective defaults, refusal of unknown and
of experimental names, the two ways to opt in."""
import os

import pytest
import torch

from spec_amd import _lib
from spec_amd.engine import Engine
from tests.test_abi import STABLE, _option_table

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


@pytest.fixture()
def no_env(monkeypatch):
    monkeypatch.delenv('SPECMI_EXPERIMENTAL', raising=False)


def test_fresh_handle_has_the_documented_defaults(no_env):
    table = _option_table(_lib.load())
    for kind in ('camcalib', 'hmr'):
        eng = Engine(kind, DEV)
        for name, (dflt, _stable) in table.items():
            assert eng.get_option(name) == dflt, (kind, name)
        for name, dflt in STABLE.items():
            assert eng.get_option(name) == dflt
        eng.close()


def test_unknown_and_experimental_names_are_refused(no_env):
    eng = Engine('hmr', DEV)
    with pytest.raises(_lib.SpecmiError) as e:
        eng.set_option('no_such_option', 1)
    assert e.value.code == _lib.ERR_ARG and 'unknown option' in str(e.value)
    with pytest.raises(_lib.SpecmiError):
        eng.set_option('no_such_float', 1.0)
    with pytest.raises(_lib.SpecmiError):
        eng.get_option('no_such_option')
    table = _option_table(_lib.load())
    for name, (dflt, stable) in table.items():
        if stable:
            continue
        eng.set_option(name, dflt)                       # the default is a no-op: always accepted
        with pytest.raises(_lib.SpecmiError) as e:
            eng.set_option(name, dflt + 1)
        assert e.value.code == _lib.ERR_STATE and 'experimental' in str(e.value), name
        assert eng.get_option(name) == dflt              # nothing changed
    for name in ('plan', 'winograd', 'fuse_downsample', 'head_collapse'):      # stable names need nothing
        eng.set_option(name, 1)
    eng.close()


def test_opt_in_by_handle_and_by_environment(no_env, monkeypatch):
    a, b = Engine('hmr', DEV), Engine('hmr', DEV)
    a.experimental()
    a.set_option('tail_fuse', 1)
    assert a.get_option('tail_fuse') == 1
    with pytest.raises(_lib.SpecmiError):
        b.set_option('tail_fuse', 1)                     # per handle: b did not opt in
    a.experimental(False)
    with pytest.raises(_lib.SpecmiError):
        a.set_option('wsplit', 0)
    monkeypatch.setenv('SPECMI_EXPERIMENTAL', '1')
    b.set_option('tail_fuse', 1)
    monkeypatch.setenv('SPECMI_EXPERIMENTAL', '0')
    with pytest.raises(_lib.SpecmiError):
        b.set_option('wsplit', 0)
    a.close(); b.close()


def test_default_path_needs_no_experimental_option(no_env):
    """With SPECMI_EXPERIMENTAL unset (tests/conftest.py sets it for the tests that pin variants and flip opt-ins): fresh modules,
    commit, the whole pipeline under every stable plan, and the reference-composed fixture - the stable surface is all the drop-in
    path itself ever touches."""
    import numpy as np
    from spec_amd import synth
    from spec_amd.pipeline import SpecPipeline
    from tests.util import golden, gpu_models, pinned_plan, rel_err, t
    assert 'SPECMI_EXPERIMENTAL' not in os.environ
    cc, hm = gpu_models(True, True, 'cuda:0')
    g = golden('hmr_e2e_camfeats.npz')
    B = int(g['batch'])
    x = t(synth.images(int(g['seed_images']), B)).to('cuda:0')
    args = [t(g[k]).to('cuda:0') for k in ('cam_rotmat', 'cam_intrinsics', 'bbox_scale', 'bbox_center', 'img_w', 'img_h')]
    for plan in ('auto', 'throughput', 'latency', 'single'):
        with pinned_plan(plan, cc, hm):
            out = hm(x, *args)
            for k in out:
                assert rel_err(out[k].cpu().numpy(), g[f'out_{k}']) < 1e-4, (plan, k)
            full = SpecPipeline(cc, hm)(x, args[2], args[3], args[4], args[5])
            assert torch.isfinite(full['smpl_vertices']).all()
    for eng in (cc.engine(torch.device('cuda:0')), hm.engine(torch.device('cuda:0'))):
        eng.set_option('winograd', 0); eng.set_option('fuse_downsample', 0)        # the stable structure switches
        eng.set_option('winograd', 1); eng.set_option('fuse_downsample', 1)
        assert eng.get_option('experimental') == 0
