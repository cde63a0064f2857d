"""spec_amd - MI355X-native implementation of the SPEC per-image inference hot path.

Public surface: ``HMR`` and ``CameraRegressorNetwork`` (drop-ins for ``spec.models.HMR`` and
``camcalib.model.CameraRegressorNetwork``), ``SpecPipeline`` (CamCalib -> decode -> SPEC fused
in-process), checkpoint helpers and the asset configuration.  All compute runs in
``lib/libspecmi.so`` (hand-written HIP for gfx950) behind the C ABI in ``include/specmi.h``.
"""
from . import assets, constants, synth  # noqa: F401


def __getattr__(name):
    # heavy imports (torch, the HIP library) are deferred until a model class is requested
    if name in ('HMR', 'CameraRegressorNetwork'):
        from . import modules
        return getattr(modules, name)
    if name in ('SpecPipeline', 'GraphedPipeline', 'AsyncGather', 'pack_outputs', 'unpack_outputs', 'gather_outputs', 'joints_payload', 'unpack_joints',
                'shard_range', 'PACKED_KEYS'):
        from . import pipeline
        return getattr(pipeline, name)
    if name == 'SPECTester':
        from . import tester
        return tester.SPECTester
    if name in ('BodyModel', 'compute_error'):
        from . import metrics
        return getattr(metrics, name)
    if name in ('load_pretrained_model', 'read_checkpoint'):
        from . import checkpoint
        return getattr(checkpoint, name)
    raise AttributeError(name)
