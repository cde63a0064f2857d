"""Import-path shim: ``from spec.utils.compute_error import compute_error`` (scripts/spec_eval.py:30 of the reference)
resolves to the MI355X build (metrics computed on the device by specmi_eval_mesh / specmi_eval_joints)."""
from spec_amd.metrics import compute_error, eval_j_24, eval_single  # noqa: F401
