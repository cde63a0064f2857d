"""Index tables must be bit-exact with the reference's spec/constants.py (golden fixture)."""
import numpy as np

from spec_amd import constants as C
from tests.util import golden


def test_joint_map_matches_reference():
    g = golden('constants.npz')
    assert np.array_equal(np.array(C.JOINT_MAP49, dtype=np.int32), g['joint_map'])
    assert list(g['joint_names']) == list(C.JOINT_NAMES49)
    assert len(C.JOINT_MAP49) == 49 and max(C.JOINT_MAP49) == 53


def test_selectors_and_norm():
    g = golden('constants.npz')
    assert np.array_equal(np.array(C.H36M_TO_J14), g['h36m_to_j14'])
    assert np.array_equal(np.array(C.J24_TO_J14), g['j24_to_j14'])
    assert np.array_equal(np.array(C.H36M_TO_J17), g['h36m_to_j17'])
    assert np.array_equal(np.array(C.J24_TO_J17), g['j24_to_j17'])
    assert np.array_equal(np.array(C.IMG_NORM_MEAN), g['img_norm_mean'])
    assert np.array_equal(np.array(C.IMG_NORM_STD), g['img_norm_std'])


def test_import_path_shim_tables():
    import spec.constants as sc
    g = golden('constants.npz')
    assert [sc.JOINT_MAP[n] for n in sc.JOINT_NAMES] == list(g['joint_map'])


def test_smpl_tables():
    assert len(C.SMPL_PARENTS) == 24 and C.SMPL_PARENTS[0] == -1
    assert all(0 <= p < i for i, p in enumerate(C.SMPL_PARENTS) if i > 0)
    assert len(C.SMPL_EXTRA_VERTEX_IDS) == 21 and max(C.SMPL_EXTRA_VERTEX_IDS) < 6890
    assert 24 + 21 + C.NUM_EXTRA_REGRESSED == 54
