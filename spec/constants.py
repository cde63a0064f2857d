"""Import-path shim: index tables / normalisation constants of the hot path."""
from spec_amd.constants import (IMG_NORM_MEAN, IMG_NORM_STD, H36M_TO_J14, H36M_TO_J17,  # noqa: F401
                                J24_TO_J14, J24_TO_J17)
from spec_amd.constants import JOINT_NAMES49 as JOINT_NAMES, JOINT_MAP49  # noqa: F401

JOINT_MAP = {}
for _n, _i in zip(JOINT_NAMES, JOINT_MAP49):
    JOINT_MAP[_n] = _i
JOINT_IDS = {JOINT_NAMES[i]: i for i in range(len(JOINT_NAMES))}
