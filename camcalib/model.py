"""Import-path shim: ``from camcalib.model import CameraRegressorNetwork`` (used at
scripts/camcalib_demo.py:32, camcalib/trainer.py:28 of the reference) resolves to the MI355X build."""
from spec_amd.modules import CameraRegressorNetwork  # noqa: F401
