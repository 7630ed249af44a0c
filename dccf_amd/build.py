# coding=utf-8
"""Builds libdccf_hip.so (gfx950) in-tree with hipcc.  `python -m dccf_amd.build` or __graft_entry__.build()."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'lib', 'libdccf_hip.so')
SRCS = sorted(glob.glob(os.path.join(HERE, 'csrc', '*.hip')))
DEPS = SRCS + sorted(glob.glob(os.path.join(HERE, 'csrc', '*.hpp'))) + [os.path.join(os.path.dirname(HERE), 'include', 'dccf_hip.h')]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in DEPS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # The host process is PyTorch-ROCm, whose wheel ships its OWN HIP runtime (torch/lib/libamdhip64.so).  Two HIP
    # runtimes in one process abort at the first launch, so the library is linked against torch's copy (same soname
    # -> the loader reuses the one torch already mapped) instead of /opt/rocm/lib/libamdhip64.so.7.
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    # -ffp-contract=off: a*b+c written as two operations stays two roundings in EVERY kernel (explicit fmaf() calls are
    # still FMAs).  The `#pragma clang fp contract(off)` in the sources says the same, but hipcc was seen to contract
    # inside a template instantiated from a header anyway (k_dense_opt_rows: s2*b2 + (1-b2)*g*g became v_pk_fma_f32
    # and the row-aware optimizer stopped matching the dense one bit for bit).
    def compile_one(src):
        obj = os.path.join(os.path.dirname(LIB), os.path.basename(src).replace('.hip', '.o'))
        c = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result', '-ffp-contract=off',
             '-c', src, '-o', obj] + os.environ.get('DCCF_EXTRA_HIPCC_FLAGS', '').split()
        if verbose:
            print(' '.join(c))
        subprocess.check_call(c)
        return obj

    # the translation units are independent: compile them side by side (the largest one, dccf_kernels.hip, takes ~95 s alone)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(len(SRCS), int(os.environ.get('DCCF_BUILD_JOBS', '0')) or (os.cpu_count() or 2) // 2))) as ex:
        objs = list(ex.map(compile_one, SRCS))
    cmd = [os.environ.get('CXX', 'g++'), '-shared', '-o', LIB] + objs + \
          ['-L' + tlib, '-lamdhip64', '-ldl', '-Wl,-rpath,' + tlib, '-Wl,--no-undefined']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
