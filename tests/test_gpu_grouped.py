"""Grouped launches: the CamCalib and SPEC trunks walked in lockstep, every layer of both as ONE launch (SURVEY.md 7 step 7's
alternative to two streams).  Same kernels, same k order: results must be bit-identical to the separate launches."""
import numpy as np
import pytest
import torch

from spec_amd import synth
from tests.util import gpu_models, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'


@pytest.mark.parametrize('B,H,W', [(1, 224, 224), (3, 224, 224), (8, 224, 224), (2, 160, 96)])
def test_trunk_pair_equals_two_trunk_calls(B, H, W):
    cc, hm = gpu_models(True, True, DEV)
    dev = torch.device(DEV)
    xa = t(synth.images(11, B)).to(dev)[:, :, :H, :W].contiguous()
    xb = t(synth.images(12, B)).to(dev)[:, :, :H, :W].contiguous()
    ea, eb = cc.engine(dev), hm.engine(dev)
    ra, rb = ea.trunk(xa).clone(), eb.trunk(xb).clone()
    ea.profile(True)
    fa, fb = ea.trunk_pair(eb, xa, xb)
    ents = ea.profile_read()
    ea.profile(False)
    assert torch.equal(fa, ra) and torch.equal(fb, rb)
    assert sum(e['launches'] for e in ents) == 2 + 16 * 3        # stem, max-pool, three fused convs per bottleneck - for BOTH trunks
    logits = ea.camcalib_head(fa)
    ref = cc(xa)
    for a_, b_ in zip(logits, ref):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize('B', [1, 5, 16])
def test_grouped_pipeline_equals_two_stream_pipeline(B):
    from spec_amd.pipeline import SpecPipeline, GraphedPipeline
    cc, hm = gpu_models(True, True, DEV)
    dev = torch.device(DEV)
    x = t(synth.images(31, B)).to(dev)
    sc, ce, iw, ih = [t(a).to(dev) for a in synth.bbox_inputs(31, B, 640., 480.)]
    from tests.util import pinned_plan
    # the launch structure never changes a bit WITHIN a plan (under 'auto' the pair and a single trunk switch plans at different
    # batch sizes - 10 and 16 images -, so the two structures may run different plans at batch 11-16)
    for plan in ('latency', 'throughput'):
        with pinned_plan(plan, cc, hm):
            ref = SpecPipeline(cc, hm, overlap=True, grouped=False)(x, sc, ce, iw, ih)
            out = SpecPipeline(cc, hm, grouped=True)(x, sc, ce, iw, ih)
            for k in ref:
                assert torch.equal(out[k], ref[k]), (plan, k)
    with pinned_plan('throughput', cc, hm):      # (restored on exit: nothing stays pinned behind this test)
        ref = SpecPipeline(cc, hm, overlap=True, grouped=False)(x, sc, ce, iw, ih)
        grp = SpecPipeline(cc, hm, grouped=True)
        gp = GraphedPipeline(grp, x, sc, ce, iw, ih)                  # one stream: captures without a side-stream fork
        out2 = gp(x, sc, ce, iw, ih)
        for k in ('smpl_vertices', 'smpl_joints2d', 'cam_vfov', 'record'):
            assert torch.equal(out2[k], ref[k]), k
        # different input shapes for the two networks: falls back to the two-stream path
        big = torch.nn.functional.interpolate(x, size=(256, 320))
        out3 = grp(x, sc, ce, iw, ih, camcalib_images=big)
        ref3 = SpecPipeline(cc, hm, overlap=True)(x, sc, ce, iw, ih, camcalib_images=big)
        assert torch.equal(out3['smpl_vertices'], ref3['smpl_vertices'])


def test_trunk_pair_resnet34_and_errors():
    from spec_amd.modules import HMR, CameraRegressorNetwork
    from spec_amd import assets
    assets.use_synthetic_assets(1003)
    dev = torch.device(DEV)
    a = CameraRegressorNetwork(backbone='resnet34').to(dev).eval()
    b = CameraRegressorNetwork(backbone='resnet34').to(dev).eval()
    g = torch.Generator().manual_seed(3)
    for m in (a, b):
        for p_ in m.parameters():
            p_.data.copy_(torch.randn(p_.shape, generator=g) * 0.05)
        for n_, buf in m.named_buffers():
            if n_.endswith('running_var'):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
    x1, x2 = t(synth.images(1, 2)).to(dev), t(synth.images(2, 2)).to(dev)
    ea, eb = a.engine(dev), b.engine(dev)
    fa, fb = ea.trunk_pair(eb, x1, x2)
    assert torch.equal(fa, ea.trunk(x1)) and torch.equal(fb, eb.trunk(x2))
    cc, _ = gpu_models(True, True, DEV)
    with pytest.raises(RuntimeError):
        ea.trunk_pair(cc.engine(dev), x1, x2)                     # resnet34 with resnet50: different depths
    with pytest.raises(ValueError):
        ea.trunk_pair(eb, x1, x2[:, :, :128])
