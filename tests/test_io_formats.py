"""Hand-off file formats: records written here are readable by the reference's
reader logic and vice versa; R/K from a record match the reference's read_cam_params fixture."""
import os

import joblib
import numpy as np
import pytest
import torch

from spec_amd import io_formats as F
from tests.util import golden


def test_camcalib_record_has_reference_layout(tmp_path):
    p = F.write_camcalib_result(str(tmp_path), '/data/imgs/frame_001.jpg', torch.tensor(0.9), torch.tensor(0.1),
                                torch.tensor(-0.05), 480)
    assert p == os.path.join(str(tmp_path), 'camcalib', 'frame_001.jpg.pkl')     # spec/utils/cam_params.py:28
    rec = joblib.load(p)
    assert set(rec) == {'vfov', 'f_pix', 'pitch', 'roll'}                         # scripts/camcalib_demo.py:135-140
    # the reference reader calls .item() on these (cam_params.py:30-33) and assigns f_pix into a tensor (:43)
    assert abs(rec['pitch'].item() - 0.1) < 1e-7 and abs(rec['roll'].item() + 0.05) < 1e-7
    t = torch.zeros(3, 3); t[0, 0] = rec['f_pix']
    assert abs(rec['f_pix'] - 480 / 2. / np.tan(np.float32(0.9) / 2.)) < 1e-3


def test_spec_result_and_eval_dump(tmp_path):
    out = {'smpl_vertices': torch.zeros(2, 6890, 3), 'pred_pose': torch.zeros(2, 24, 3, 3),
           'pred_shape': torch.zeros(2, 10), 'pred_cam': torch.ones(2, 3)}
    p = F.write_spec_result(str(tmp_path), '/x/img_07.png', out)
    assert p.endswith(os.path.join('spec_results', 'img_07.pkl'))                 # spec/tester.py:158-162
    back = joblib.load(p)
    assert all(isinstance(v, np.ndarray) for v in back.values()) and back['smpl_vertices'].shape == (2, 6890, 3)
    d = F.EvalDump(); d.add(out, imgnames=['a.jpg', 'b.jpg'], dataset_name='spec-syn'); d.add(out, imgnames=['c.jpg', 'd.jpg'], dataset_name='spec-syn')
    q = d.write(str(tmp_path), 'spec-syn')
    assert os.path.basename(q) == 'evaluation_results_spec-syn.pkl'               # spec/trainer.py:533-536
    ev = joblib.load(q)
    # the reference's evaluation_results keys (spec/trainer.py:118-135); compute_error reads ['vertices'] (compute_error.py:108)
    assert ev['vertices'].shape == (4, 6890, 3) and ev['pose'].shape == (4, 24, 3, 3)
    assert set(ev) == {'pose', 'shape', 'cam', 'vertices', 'imgname', 'dataset_name'}
    assert list(ev['imgname']) == ['a.jpg', 'b.jpg', 'c.jpg', 'd.jpg']


@pytest.mark.gpu
def test_read_cam_params_matches_reference_fixture(tmp_path):
    g = golden('cam_params.npz')
    for i, (pitch, roll, vfov, h, w, f_pix) in enumerate(g['meta']):
        os.makedirs(os.path.join(str(tmp_path), 'camcalib'), exist_ok=True)
        joblib.dump({'vfov': np.float32(vfov), 'f_pix': np.float64(f_pix), 'pitch': np.float32(pitch),
                     'roll': np.float32(roll)}, os.path.join(str(tmp_path), 'camcalib', f'im{i}.jpg.pkl'))
        R, K, v, p, r, f = F.read_cam_params(str(tmp_path), f'/x/im{i}.jpg', (int(h), int(w)))
        assert np.abs(R.cpu().numpy() - g['R'][i]).max() < 2e-6
        assert np.array_equal(K.cpu().numpy(), g['K'][i])
        assert abs(v - np.float32(vfov)) < 1e-7 and abs(p - np.float32(pitch)) < 1e-7 and f == f_pix
