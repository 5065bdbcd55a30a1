"""In-tree build of libhrfuser_hip.so (gfx950) with hipcc.  No JIT cache: the .so sits next to
the sources so it travels to the GPU box with the repo snapshot.

    python -m hrfuser_amd.build_ext            # build if stale
    python -m hrfuser_amd.build_ext --force
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, 'libhrfuser_hip.so')
SOURCES = ['conv_engine.hip', 'lin_engine.hip', 'conv3_engine.hip', 'conv3w_engine.hip', 'dwconv.hip', 'attention.hip', 'pointwise.hip', 'replay.hip']
HEADERS = ['hrf_rt.h', 'hrf_common.h', 'hrf_lin.h', 'hrf_replay.h', os.path.join(ROOT, 'include', 'hrfuser_hip.h')]
ARCH = 'gfx950'


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        p = f if os.path.isabs(f) else os.path.join(CSRC, f)
        if os.path.exists(p):
            with open(p, 'rb') as fh:
                h.update(fh.read())
    return h.hexdigest()


def _torch_lib_dir():
    try:
        import torch
        return os.path.join(os.path.dirname(torch.__file__), 'lib')
    except Exception:
        return None


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the C-ABI shared library."""
    stamp = LIB + '.stamp'
    dg = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dg:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [hipcc, '-x', 'hip', f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC',
               '-ffp-contract=off', '-c', sp, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
    # Link against the HIP runtime by its unversioned soname: inside a PyTorch-ROCm process the
    # already-loaded libamdhip64.so (torch/lib) satisfies it, so the kernels share torch's
    # runtime, streams and allocations; standalone it resolves through the ROCm library path.
    tl = _torch_lib_dir()
    libdir = tl if tl and os.path.exists(os.path.join(tl, 'libamdhip64.so')) else '/opt/rocm/lib'
    cmd = ['g++', '-shared', '-o', LIB] + objs + [f'-L{libdir}', '-lamdhip64',
                                                  '-Wl,-rpath,/opt/rocm/lib']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, 'w') as fh:
        fh.write(dg)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
