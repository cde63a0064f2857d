#!/usr/bin/env python
"""Golden vectors for the arg-max ('kl' / 'ce' / legacy) decode of CamCalib and the import surface of the
drop-in modules -- run ONLY in the build container (needs /root/reference).

(1) ``cam_bins.npz``: the bin tables of the reference's ``camcalib/cam_utils.py:23-63`` (scipy-built roll table
    included), seeded logits with ties / plateaus / +-inf, and what the reference's own ``bins2vfov / bins2pitch /
    bins2roll / bins2horizon`` and ``convert_preds_to_angles(loss_type='kl')`` return for them.
(2) ``import_surface.json``: every name the reference's own files import from the modules the build replaces
    (``spec.models``, ``spec.models.hmr``, ``spec.constants``, ``camcalib.model``, ``camcalib.cam_utils``,
    ``spec.utils.cam_params``, ``spec.utils.compute_error``, ``spec.tester``), found by parsing the reference with ``ast`` (import
    statements and ``constants.X`` attribute uses).  Only names and arrays are stored, no reference source.
"""
import ast
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'

from spec_amd import synth  # noqa: E402
from oracle import refshim  # noqa: E402

TARGETS = {'spec.models', 'spec.models.hmr', 'spec.constants', 'camcalib.model', 'camcalib.cam_utils',
           'spec.utils.cam_params', 'spec.utils.compute_error', 'spec.tester'}


def resolve(module, level, pkg):
    if level == 0:
        return module
    parts = pkg.split('.')
    base = parts[:len(parts) - (level - 1)]
    return '.'.join(base + ([module] if module else []))


def scan():
    surface = {t: set() for t in TARGETS}
    users = {}
    for top in ('spec', 'camcalib', 'scripts'):
        for dp, _, files in os.walk(os.path.join(REF, top)):
            for fn in files:
                if not fn.endswith('.py'):
                    continue
                path = os.path.join(dp, fn)
                rel = os.path.relpath(path, REF)
                pkg = os.path.dirname(rel).replace(os.sep, '.')
                tree = ast.parse(open(path).read())
                aliases = {}          # local name -> target module (e.g. `from . import constants`)
                for node in ast.walk(tree):
                    if isinstance(node, ast.ImportFrom):
                        mod = resolve(node.module, node.level, pkg)
                        for a in node.names:
                            full = f'{mod}.{a.name}' if mod else a.name
                            if mod in TARGETS and a.name != '*':
                                surface[mod].add(a.name)
                                users.setdefault(f'{mod}.{a.name}', []).append(f'{rel}:{node.lineno}')
                            if full in TARGETS:
                                aliases[a.asname or a.name] = full
                for node in ast.walk(tree):
                    if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in aliases:
                        mod = aliases[node.value.id]
                        surface[mod].add(node.attr)
                        users.setdefault(f'{mod}.{node.attr}', []).append(f'{rel}:{node.lineno}')
    return {k: sorted(v) for k, v in surface.items()}, {k: sorted(set(v)) for k, v in users.items()}


def main():
    ref = refshim.import_reference()
    CU = ref['cam_utils']
    nb = 256
    rows = []
    r = np.zeros(nb, np.float32); rows.append(r)                                   # all equal -> index 0
    r = np.full(nb, -1.0, np.float32); r[[17, 99, 255]] = 2.5; rows.append(r)      # three-way tie -> 17
    r = np.full(nb, -np.inf, np.float32); r[255] = 0.0; rows.append(r)            # last bin
    r = np.full(nb, 3.0, np.float32); r[0] = np.inf; rows.append(r)                # +inf
    r = np.linspace(-1, 1, nb).astype(np.float32); r[200:] = r[200]; rows.append(r)   # plateau at the top -> 200
    r = np.full(nb, -0.0, np.float32); r[5] = 0.0; rows.append(r)                  # -0.0 == +0.0 -> 0
    rnd = synth.normal(91, 'bins.logits', (58, nb), std=2.0)
    rnd = np.round(rnd * 4) / 4                                                    # quarter steps: many exact ties
    logits = np.concatenate([np.stack(rows), rnd], 0).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    lv, lp, lr = logits, np.roll(logits, 5, axis=0), np.roll(logits, 11, axis=0)
    kv, kp, kr = CU.convert_preds_to_angles(t(lv), t(lp), t(lr))                   # default loss_type='kl', torch
    assert kv.dtype == torch.float64
    nv, npi, nr = CU.convert_preds_to_angles(t(lv), t(lp), t(lr), loss_type='ce', return_type='np')
    assert np.array_equal(kv.numpy(), nv) and np.array_equal(kr.numpy(), nr) and np.array_equal(kp.numpy(), npi)
    lg = CU.convert_preds_to_angles(t(lv), t(lp), t(lr), loss_type='softargmax_l2', legacy=True)
    np.savez(os.path.join(OUT, 'cam_bins.npz'),
             logits_vfov=lv, logits_pitch=lp, logits_roll=lr,
             kl_vfov=kv.numpy(), kl_pitch=kp.numpy(), kl_roll=kr.numpy(),
             horizon=CU.bins2horizon(t(lv)), bins3d_pitch=CU.bins2pitch(t(logits.reshape(4, 16, nb))),
             legacy_vfov=lg[0].numpy(), legacy_pitch=lg[1].numpy(), legacy_roll=np.asarray(lg[2]),
             softargmax=CU.get_softargmax(t(lv)).numpy(),
             **{n: getattr(CU, n) for n in ('pitch_bins', 'pitch_bins_centers', 'horizon_bins', 'horizon_bins_centers',
                                            'roll_bins', 'roll_bins_centers', 'vfov_bins', 'vfov_bins_centers',
                                            'roll_new_bins', 'roll_new_bins_centers')},
             soft_idx=np.stack([CU.vfov2soft_idx(np.linspace(0.3, 2.0, 7)), CU.pitch2soft_idx(np.linspace(-0.5, 0.5, 7)),
                                CU.roll2soft_idx(np.linspace(-0.5, 0.5, 7))]))
    surface, users = scan()
    with open(os.path.join(OUT, 'import_surface.json'), 'w') as f:
        json.dump({'names': surface, 'used_at': users}, f, indent=1, sort_keys=True)
    print({k: len(v) for k, v in surface.items()})
    for k, v in surface.items():
        print(k, v)


if __name__ == '__main__':
    main()
