"""The one-command upstream pin (tests/golden/make_fixtures.py --upstream / --selfcheck), exercised here with the shim -
and with a throw-away ``pare`` / ``smplx`` / ``loguru`` package tree that re-exports the oracle's leaves - playing the
upstream.  Needs the reference checkout, i.e. runs in the build container only (skipped elsewhere)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, 'tests', 'golden', 'make_fixtures.py')
pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/spec'), reason='reference checkout not present')


def _run(args, extra_path=None):
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join(([extra_path] if extra_path else []) + [ROOT, env.get('PYTHONPATH', '')])
    return subprocess.run([sys.executable, SCRIPT] + args, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.timeout(700)
def test_selfcheck_reproduces_committed_fixtures_bit_for_bit():
    r = _run(['--selfcheck'])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'worst relative deviation over all arrays: 0.000e+00' in r.stdout
    assert "'pare': 'shim'" in r.stdout


def _write_fake_upstream(root):
    files = {
        'pare/__init__.py': '',
        'pare/models/__init__.py': 'SMPL = None\n',
        'pare/models/backbone/__init__.py': 'from oracle.resnet import resnet50\n__all__ = ["resnet50"]\n',
        'pare/models/backbone/utils.py': 'from oracle.resnet import get_backbone_info\n',
        'pare/models/backbone/hrnet.py': 'hrnet_w32 = None\nhrnet_w48 = None\n',
        'pare/models/head/__init__.py': 'from oracle.heads import HMRHead, SMPLHead, SMPLCamHead\n',
        'pare/models/layers/__init__.py': '',
        'pare/models/layers/softargmax.py': 'from oracle.geometry import softargmax1d\n',
        'pare/utils/__init__.py': '',
        'pare/utils/train_utils.py': 'def load_pretrained_model(*a, **k):\n    return None\n',
        'pare/utils/geometry.py': 'from oracle.geometry import batch_euler2matrix, rot6d_to_rotmat, rotmat_to_rot6d\n',
        'pare/utils/eval_utils.py': 'from oracle.metrics import compute_error_verts, reconstruction_error\n',
        'pare/core/__init__.py': '',
        'pare/core/constants.py': 'from oracle.metrics import H36M_TO_J14 as _h\nH36M_TO_J14 = list(_h)\n',
        'smplx/__init__.py': 'SMPL = object\n',
        'loguru/__init__.py': textwrap.dedent('''
            class _L:
                def __getattr__(self, name):
                    return lambda *a, **k: None
            logger = _L()
        '''),
    }
    for rel, body in files.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            f.write(body)


@pytest.mark.timeout(700)
def test_upstream_mode_binds_real_packages_when_they_import(tmp_path):
    """Installed leaf packages must be used INSTEAD of the shim (here: a fake upstream that re-exports the oracle, so the
    regenerated fixtures must still equal the committed ones), and the report must say so."""
    import json
    _write_fake_upstream(str(tmp_path))
    rep_path = str(tmp_path / 'pin.json')
    r = _run(['--upstream', '--report', rep_path], extra_path=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "'pare': 'upstream'" in r.stdout and "'smplx': 'upstream'" in r.stdout and "'loguru': 'upstream'" in r.stdout
    assert 'worst relative deviation over all arrays: 0.000e+00' in r.stdout
    rep = json.load(open(rep_path))          # the machine-checkable form of the same run
    assert rep['pass'] is True and rep['mode'] == 'upstream' and rep['pinned_upstream'] == ['loguru', 'pare', 'smplx']
    assert rep['tolerance_fp_rel'] == 1e-5 and rep['worst_rel_dev'] == 0.0 and not rep['problems']
    assert len(rep['arrays']) > 40 and all(a['max_abs_dev'] == 0.0 for a in rep['arrays'])
    assert {'pare', 'smplx', 'torch', 'numpy', 'opencv-python'} <= set(rep['packages'])
    assert any(a['exact_required'] for a in rep['arrays'])           # index tables are held to byte equality


@pytest.mark.timeout(700)
def test_upstream_report_fails_above_the_tolerance(tmp_path):
    """A leaf that disagrees with the restatement by more than 1e-5 must turn the exit code and the report red: here the fake
    upstream's rot6d_to_rotmat is perturbed by 1e-3."""
    import json
    _write_fake_upstream(str(tmp_path))
    with open(os.path.join(str(tmp_path), 'pare/utils/geometry.py'), 'w') as f:
        f.write('from oracle.geometry import batch_euler2matrix, rotmat_to_rot6d\n'
                'from oracle import geometry as _g\n'
                'def rot6d_to_rotmat(x):\n    return _g.rot6d_to_rotmat(x) * (1.0 + 1e-3)\n')
    with open(os.path.join(str(tmp_path), 'pare/models/head/__init__.py'), 'w') as f:
        f.write('from oracle import heads as _h\nimport pare.utils.geometry as _pg\n'
                '_h.rot6d_to_rotmat = _pg.rot6d_to_rotmat\nfrom oracle.heads import HMRHead, SMPLHead, SMPLCamHead\n')
    rep_path = str(tmp_path / 'pin.json')
    r = _run(['--upstream', '--report', rep_path], extra_path=str(tmp_path))
    rep = json.load(open(rep_path))
    assert r.returncode == 1 and rep['pass'] is False and rep['worst_rel_dev'] > 1e-5, (r.returncode, rep['worst_rel_dev'])


@pytest.mark.timeout(700)
def test_upstream_mode_without_the_packages_says_so():
    r = _run(['--upstream'])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'none of pare / smplx / loguru is importable' in r.stdout
