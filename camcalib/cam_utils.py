"""Import-path shim for ``camcalib.cam_utils`` of the reference (used at scripts/camcalib_demo.py:34,179-180,
camcalib/trainer.py:31, camcalib/pano_dataset.py:31-32): every public name resolves to the MI355X build
(spec_amd.cam_utils: bin tables on the host, reductions over the bins on the GPU)."""
from spec_amd.cam_utils import (  # noqa: F401
    get_bins, pitch_bins, pitch_bins_centers, horizon_bins, horizon_bins_centers, roll_bins, roll_bins_centers,
    vfov_bins, vfov_bins_centers, roll_new_bins, roll_new_bins_centers,
    bins2horizon, bins2pitch, bins2roll, bins2vfov, vfov2soft_idx, pitch2soft_idx, roll2soft_idx,
    angle_to_soft_idx, soft_idx_to_angle, get_softargmax, convert_preds_to_angles,
    decode_camera, cam_params_from_angles)
