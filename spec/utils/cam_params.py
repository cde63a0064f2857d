"""Import-path shim: ``from spec.utils.cam_params import read_cam_params`` (spec/tester.py:34 of the reference)
resolves to the MI355X build (R / K built on the device by specmi_cam_params)."""
from spec_amd.io_formats import read_cam_params  # noqa: F401
