"""Oracle (test infrastructure): name shim that lets the reference's OWN in-tree modules
(``spec/models/hmr.py``, ``camcalib/model.py``, ``camcalib/cam_utils.py``,
``spec/utils/cam_params.py``, ``spec/constants.py``) import in the build container, where
``pare``, ``smplx``, ``loguru`` ... are not installed.  The missing names are bound to this
oracle's restatements, so running the reference modules pins the *composition* of the path
(kwarg routing, formulae in the in-tree files, dict keys) - not the leaf arithmetic.

Used ONLY by ``tests/golden/make_fixtures.py`` in the build container; ``/root/reference``
does not exist on the GPU box and nothing at test/bench time imports this module.
"""
import importlib
import importlib.util
import sys
import types


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def have(pkg):
    """True when the REAL package is importable (and is not one of this shim's stub modules)."""
    m = sys.modules.get(pkg)
    if m is not None:
        return getattr(m, '__file__', None) is not None
    try:
        return importlib.util.find_spec(pkg) is not None
    except (ImportError, ValueError):
        return False


BOUND = {}   # filled by install(): leaf package -> 'upstream' | 'shim'


def install(reference_root='/root/reference', upstream=False):
    """``upstream=True``: wherever the reference's real leaf packages import (``pare`` needs ``smplx`` and ``loguru``
    too), they are LEFT ALONE and the reference's modules bind to them; only what is missing is stubbed.  The
    fixtures regenerated that way pin the upstream leaf arithmetic itself (``tests/golden/make_fixtures.py --upstream``)."""
    from . import geometry, heads, resnet
    BOUND.clear()
    real_pare = bool(upstream) and have('pare') and have('smplx')
    BOUND['pare'] = 'upstream' if real_pare else 'shim'
    BOUND['smplx'] = 'upstream' if real_pare else 'shim'

    class _Logger:
        def __getattr__(self, _name):
            return lambda *a, **k: None

    if upstream and have('loguru'):
        BOUND['loguru'] = 'upstream'
    else:
        BOUND['loguru'] = 'shim'
        _mod('loguru', logger=_Logger())
    joblib_stub = sys.modules.get('joblib')
    if joblib_stub is None:
        try:
            import joblib  # noqa: F401
        except Exception:
            _mod('joblib', load=lambda *a, **k: None, dump=lambda *a, **k: None)

    if real_pare:
        if reference_root not in sys.path:
            sys.path.insert(0, reference_root)
        return BOUND
    pare = _mod('pare')
    models = _mod('pare.models', SMPL=None)
    backbone = _mod('pare.models.backbone', resnet50=resnet.resnet50)
    backbone.__all__ = ['resnet50']
    butils = _mod('pare.models.backbone.utils', get_backbone_info=resnet.get_backbone_info)
    hrnet = _mod('pare.models.backbone.hrnet', hrnet_w32=None, hrnet_w48=None)
    head = _mod('pare.models.head', HMRHead=heads.HMRHead, SMPLHead=heads.SMPLHead,
                SMPLCamHead=heads.SMPLCamHead)
    layers = _mod('pare.models.layers')
    softargmax = _mod('pare.models.layers.softargmax', softargmax1d=geometry.softargmax1d)
    utils = _mod('pare.utils')
    train_utils = _mod('pare.utils.train_utils', load_pretrained_model=lambda *a, **k: None)
    from . import metrics as _metrics
    _mod('pare.utils.eval_utils', compute_error_verts=_metrics.compute_error_verts,
         reconstruction_error=_metrics.reconstruction_error)
    _mod('pare.core')
    _mod('pare.core.constants', H36M_TO_J14=list(_metrics.H36M_TO_J14))
    _mod('smplx', SMPL=object)
    try:
        import tqdm  # noqa: F401
    except Exception:
        _mod('tqdm', tqdm=lambda x, **k: x)
    geom = _mod('pare.utils.geometry', batch_euler2matrix=geometry.batch_euler2matrix,
                rot6d_to_rotmat=geometry.rot6d_to_rotmat, rotmat_to_rot6d=geometry.rotmat_to_rot6d)
    pare.models, pare.utils = models, utils
    models.backbone, models.head, models.layers = backbone, head, layers
    backbone.utils, backbone.hrnet = butils, hrnet
    layers.softargmax = softargmax
    utils.train_utils, utils.geometry = train_utils, geom
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    return BOUND


def import_reference(reference_root='/root/reference', upstream=False):
    """Returns the reference's modules (hmr, camcalib.model, cam_utils, cam_params, constants)."""
    install(reference_root, upstream=upstream)
    # the build tree also has `spec` / `camcalib` import-path packages; make sure the
    # reference's win for this process.
    for name in [n for n in sys.modules if n == 'spec' or n.startswith('spec.')
                 or n == 'camcalib' or n.startswith('camcalib.')]:
        del sys.modules[name]
    sys.path = [reference_root] + [p for p in sys.path if p != reference_root]
    mods = {}
    mods['hmr'] = importlib.import_module('spec.models.hmr')
    mods['camcalib_model'] = importlib.import_module('camcalib.model')
    mods['cam_utils'] = importlib.import_module('camcalib.cam_utils')
    mods['cam_params'] = importlib.import_module('spec.utils.cam_params')
    mods['constants'] = importlib.import_module('spec.constants')
    # spec/utils/compute_error.py does `from ..config import ...` (yacs-based, not importable here)
    if not (upstream and have('yacs')):
        _mod('spec.config', DATASET_FILES=[{}, {}], SMPL_MODEL_DIR='')
    mods['compute_error'] = importlib.import_module('spec.utils.compute_error')
    for m in mods.values():
        assert m.__file__.startswith(reference_root), m.__file__
    return mods
