"""Concurrency contract of include/specmi.h:31-36 as round 5 stated it: a handle is driven from one thread and one stream at a
time, distinct handles are independent - so worker threads, each with its OWN modules on its OWN ``torch.cuda.Stream``, may run
forwards and the ``cam_utils`` helpers (per-thread decode engines) concurrently.  The reference is single-threaded on the default
stream (spec/tester.py:90-163); this pins what the drop-in adds for a serving loop."""
import gc
import threading

import numpy as np
import pytest
import torch

from spec_amd import cam_utils, synth
from tests.util import synth_states, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'


def _fresh_models():
    from spec_amd import assets
    from spec_amd.modules import HMR, CameraRegressorNetwork
    assets.use_synthetic_assets(1003)
    cs, hs = synth_states(True)
    cc = CameraRegressorNetwork()
    cc.load_state_dict({k: t(v) for k, v in cs.items()}, strict=True)
    hm = HMR(use_cam=True, use_cam_feats=True)
    hm.load_state_dict({k: t(v) for k, v in hs.items()}, strict=False)
    return cc.to(DEV).eval(), hm.to(DEV).eval()


def _work(cc, hm, inputs, iters, results, errors, tag, barrier=None):
    """``iters`` rounds of: CamCalib forward -> bins2* / soft-arg-max helpers -> decode -> HMR forward, on the thread's stream."""
    try:
        stream = torch.cuda.Stream(device=DEV)
        outs = []
        with torch.cuda.stream(stream):
            if barrier is not None:
                barrier.wait()
            for i in range(iters):
                x, sc, ce, iw, ih = inputs[i % len(inputs)]
                lg = cc(x)
                vf_bins = cam_utils.bins2vfov(lg[0])                         # host round trip (arg-max on the device)
                soft = cam_utils.get_softargmax(lg[1])
                cam = cam_utils.decode_camera(*lg, img_h=ih, img_w=iw)
                out = hm(x, cam['cam_rotmat'], cam['cam_intrinsics'], sc, ce, iw, ih)
                if i >= iters - len(inputs):                                   # keep the last pass over the inputs
                    outs.append({'lg0': lg[0].clone(), 'vf_bins': np.asarray(vf_bins).copy(), 'soft': soft.clone(),
                                 'vfov': cam['vfov'].clone(), 'verts': out['smpl_vertices'].clone(), 'j2d': out['smpl_joints2d'].clone()})
            stream.synchronize()
        results[tag] = (outs, cc.engine(torch.device(DEV)).sync_status(), hm.engine(torch.device(DEV)).sync_status())
    except BaseException as e:      # noqa: BLE001 - surfaced by the test
        errors[tag] = e


def test_two_threads_two_streams_equal_serial():
    iters = 200
    models = [_fresh_models() for _ in range(2)]
    inputs = []
    for j, B in enumerate((1, 2, 3, 8)):                                       # single / latency plans: the ones with in-launch hand-offs
        x = t(synth.images(500 + j, B)).to(DEV)
        sc, ce, iw, ih = [t(a).to(DEV) for a in synth.bbox_inputs(500 + j, B, 640., 480.)]
        inputs.append((x, sc, ce, iw, ih))
    torch.cuda.synchronize()
    serial, errors = {}, {}
    _work(*models[0], inputs, len(inputs), serial, errors, 'serial')
    assert not errors, errors
    ref = serial['serial'][0]
    results = {}
    barrier = threading.Barrier(2)
    threads = [threading.Thread(target=_work, args=(*models[k], inputs, iters, results, errors, k, barrier)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
        assert not th.is_alive(), 'worker thread hung'
    assert not errors, errors
    for k in range(2):
        outs, st_cc, st_hm = results[k]
        assert st_cc == 0 and st_hm == 0, (k, st_cc, st_hm)
        assert len(outs) == len(ref)
        for a, b in zip(outs, ref):
            for key in a:
                same = np.array_equal(a[key], b[key]) if isinstance(a[key], np.ndarray) else torch.equal(a[key], b[key])
                assert same, (k, key)


def test_decode_engines_are_per_thread_and_die_with_it():
    """cam_utils keeps its parameter-less decode engine in thread-local storage: two threads get two handles, a thread's handle is
    destroyed when the thread ends (nothing accumulates in a module-level table)."""
    import weakref
    seen, refs = {}, []
    both = threading.Barrier(2)

    errs = []

    def grab(tag):
        try:
            e = cam_utils._engine(torch.device(DEV))
            seen[tag] = (id(e), e.h.value)
            refs.append(weakref.ref(e))
            both.wait(timeout=60)                          # both engines exist at the same time: distinct objects, distinct handles
            x = torch.randn(4, 256, device=DEV)
            assert np.array_equal(cam_utils.bins2pitch(x), cam_utils.pitch_bins_centers[x.argmax(-1).cpu().numpy()])
        except BaseException as ex:     # noqa: BLE001
            errs.append(ex)

    ths = [threading.Thread(target=grab, args=(k,)) for k in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    mine = cam_utils._engine(torch.device(DEV))
    assert seen[0][0] != seen[1][0] and seen[0][1] != seen[1][1]
    gc.collect()
    assert all(r() is None for r in refs), 'decode engines of finished threads are still alive'
    assert mine is cam_utils._engine(torch.device(DEV))      # the calling thread keeps its own
