"""Import-path shim: ``from spec.models.hmr import HMR`` resolves to the MI355X build."""
from spec_amd.modules import HMR  # noqa: F401
