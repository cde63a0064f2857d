"""Round 5, host side (no GPU): the small-batch floor model of bench.py, the one place that holds the launch-structure rule, the
new C-ABI entry points declared / exported / bound."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_resnet50_layer_list_matches_the_survey_flop_count():
    """53 convolutions whose MACs add up to SURVEY.md 8d's 4.087 GMAC per trunk at 224 x 224 (the figure `value` is priced with)."""
    b = _bench()
    layers = b.resnet50_layers()
    assert len(layers) == 53
    gflop = sum(2.0 * cin * cout * k * k * o * o for _, cin, cout, k, _s, o in layers) / 1e9
    assert abs(gflop - b.TRUNK_GFLOP_PER_IMAGE) < 1e-6
    assert sum(1 for n, *_ in layers if n.endswith('downsample')) == 4


def test_small_batch_floor_is_a_lower_bound_model():
    b = _bench()
    prev = 0.0
    for batch in (1, 2, 4, 8):
        f = b.small_batch_floor(batch, 58)
        assert f['floor_ms'] > prev and f['floor_with_boundaries_ms'] > f['floor_ms']
        # never below the pure MFMA time of the un-padded MACs nor below one read of all weights
        assert f['floor_ms'] >= batch * 2 * b.TRUNK_GFLOP_PER_IMAGE / b.PEAK_FP32_MFMA_TFLOPS * 1e-3 * 0.999
        assert f['floor_ms'] >= f['weight_stream_part_ms'] * 0.999
        prev = f['floor_ms']
    # batch 1 pays for the rows it pads to the 32-row MFMA tile (layer4: 49 -> 64): more than 1/8 of the batch-8 figure
    assert b.small_batch_floor(1, 0)['mfma_part_ms'] > b.small_batch_floor(8, 0)['mfma_part_ms'] / 8


def test_launch_structure_rule_lives_in_one_place():
    from spec_amd.pipeline import SpecPipeline

    class Dummy:
        use_cam = True
    pipe = SpecPipeline(Dummy(), Dummy())
    assert [n for n in range(1, 30) if SpecPipeline.auto_groups(n)] == [1, 2, 3, 17, 18, 19, 20]
    for n in (1, 3, 4, 10, 11, 16, 17, 20, 21):
        st = pipe.launch_structure((n, 3, 224, 224))
        assert st['grouped'] == SpecPipeline.auto_groups(n)
        assert ('grouped launch' in st['structure']) == st['grouped']
        assert st['plan'] is None                     # no engine behind the dummies: the library is not asked
    # CamCalib on another input shape (the demo's full frame) can never be grouped
    assert not pipe.launch_structure((1, 3, 224, 224), (1, 3, 600, 1066))['grouped']
    # bench.py must not hold a copy of the rule
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert 'launch_structure' in src and not re.search(r'1[17]\s*<=\s*b\s*<=\s*(16|20)', src)


def test_round5_entry_points_are_declared_and_bound():
    from spec_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'specmi.h')).read()
    for name in ('specmi_trunk_plan', 'specmi_sync_status', 'specmi_sync_reset', 'specmi_debug_poison_sync', 'specmi_camcalib_head_decode'):
        assert re.search(r'\bint\s+' + name + r'\s*\(', hdr), name
        assert name in _lib.PROTOTYPES, name
    lib = _lib.load()
    for name in _lib.PROTOTYPES:
        assert hasattr(lib, name), name
    # the options a caller can reach are documented where the ABI is
    for opt in ('"wsplit"', '"persist"', '"tail_fuse"', '"single_max_batch"', '3 = single'):
        assert opt in hdr, opt
