"""Import-path shim for the CamCalib decode helpers (GPU implementation in spec_amd.cam_utils)."""
from spec_amd.cam_utils import (convert_preds_to_angles, decode_camera, soft_idx_to_angle,  # noqa: F401
                                angle_to_soft_idx)
