from spec_amd.modules import HMR  # drop-in for the reference import path spec.models.HMR (spec/models/__init__.py:1)
