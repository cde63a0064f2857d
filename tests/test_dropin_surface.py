"""The drop-in import surface: every name the reference's own files import from the modules this build replaces
(tests/golden/import_surface.json, produced by tests/golden/make_bins_fixture.py scanning /root/reference with ast)
resolves against the build's shim packages, and the host-side bin tables equal the reference's bit for bit."""
import importlib
import inspect
import json
import os

import numpy as np
import pytest

from tests.util import GOLDEN, golden


def _surface():
    with open(os.path.join(GOLDEN, 'import_surface.json')) as f:
        return json.load(f)


def test_every_imported_name_resolves():
    s = _surface()
    assert len(s['names']['camcalib.cam_utils']) >= 15
    missing = []
    for mod, names in s['names'].items():
        m = importlib.import_module(mod)
        for n in names:
            if not hasattr(m, n):
                missing.append(f'{mod}.{n}  (used at {s["used_at"].get(mod + "." + n)})')
    assert not missing, missing


def test_demo_import_lines_work():
    # scripts/camcalib_demo.py:32,34,179-180 / camcalib/trainer.py:28,31 / spec/tester.py:32,34 of the reference
    from camcalib.model import CameraRegressorNetwork  # noqa: F401
    from camcalib.cam_utils import bins2vfov, bins2pitch, bins2roll, convert_preds_to_angles  # noqa: F401
    from camcalib.cam_utils import roll_new_bins_centers as roll_bins_centers  # noqa: F401
    from camcalib.cam_utils import pitch_bins_centers, vfov_bins_centers  # noqa: F401
    from spec.models import HMR  # noqa: F401
    from spec.utils.cam_params import read_cam_params  # noqa: F401
    from spec.utils.compute_error import compute_error  # noqa: F401
    sig = inspect.signature(convert_preds_to_angles)
    assert sig.parameters['loss_type'].default == 'kl'            # camcalib/cam_utils.py:122
    assert sig.parameters['return_type'].default == 'torch' and sig.parameters['legacy'].default is False


def test_bin_tables_bit_exact_vs_reference():
    from spec_amd import cam_utils as CU
    g = golden('cam_bins.npz')
    for n in ('pitch_bins', 'pitch_bins_centers', 'horizon_bins', 'horizon_bins_centers', 'roll_bins',
              'roll_bins_centers', 'vfov_bins', 'vfov_bins_centers', 'roll_new_bins', 'roll_new_bins_centers'):
        a = getattr(CU, n)
        assert a.dtype == np.float64 and a.shape == g[n].shape
        assert np.array_equal(a, g[n]), n                         # roll_bins: scipy's norm.pdf restated in NumPy
    si = np.stack([CU.vfov2soft_idx(np.linspace(0.3, 2.0, 7)), CU.pitch2soft_idx(np.linspace(-0.5, 0.5, 7)),
                   CU.roll2soft_idx(np.linspace(-0.5, 0.5, 7))])
    assert np.array_equal(si, g['soft_idx'])
    assert CU.soft_idx_to_angle(0.25, 0.2617, 2.1) == (2.1 - 0.2617) * ((0.25 + 1) / 2) + 0.2617


def test_argmax_decode_needs_gpu_not_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from spec_amd import cam_utils as CU
    with pytest.raises(Exception):
        CU.bins2vfov(torch.zeros(2, 256))
