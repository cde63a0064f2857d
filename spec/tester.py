"""Import-path shim: ``from spec.tester import SPECTester`` (scripts/spec_demo.py:28 of the reference) resolves to the
MI355X build."""
from spec_amd.tester import SPECTester  # noqa: F401
