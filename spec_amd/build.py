"""Build libspecmi.so (HIP, gfx950 only) in-tree: spec_amd/lib/libspecmi.so.

hipcc cross-compiles without a GPU.  The library is git-ignored but travels to the GPU box
with the working-tree snapshot.  ``python -m spec_amd.build`` or ``build()`` from Python.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libspecmi.so')
SOURCES = ['api.hip', 'commit.hip', 'options.hip', 'hrnet.hip', 'conv_igemm.hip', 'conv_persist.hip', 'conv_wsplit.hip', 'conv_wino.hip', 'conv_bf16s.hip', 'stem.hip', 'head.hip', 'smpl.hip', 'eval.hip', 'preprocess.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _hipcc() -> str:
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, 'specmi_internal.h'), os.path.join(CSRC, 'handle.h'), os.path.join(CSRC, 'conv_igemm_tile.h'),
               os.path.join(CSRC, 'conv_igemm_body.inc'),
               os.path.join(os.path.dirname(HERE), 'include', 'specmi.h')]

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return o, True
        return o, False

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
