"""Build the CPU-emulation library of the kernels (TEST INFRASTRUCTURE, never loaded by the
product).  Same sources as libhrfuser_hip.so, compiled by g++ with -DHRF_EMUL."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'hrfuser_amd', 'csrc')
OUT = os.path.join(HERE, '_build')
LIB = os.path.join(OUT, 'libhrfuser_emul.so')
SOURCES = ['conv_engine.hip', 'wgrad_tiled.hip', 'lin_engine.hip', 'lin2_engine.hip', 'conv3_engine.hip', 'conv3x_engine.hip', 'wgrad3x_engine.hip', 'conv3w_engine.hip', 'dwconv.hip', 'attention.hip', 'attn_block.hip', 'ffn_eval.hip', 'pointwise.hip', 'group.hip', 'p2p_exchange.hip']


def build(force=False, sanitize=False):
    sanitize = sanitize or os.environ.get('HRF_EMUL_ASAN') == '1'      # sanitizer runs: HRF_EMUL_ASAN=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 pytest ...
    os.makedirs(OUT, exist_ok=True)
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))]
    files += [os.path.join(HERE, 'emul_rt.h'), os.path.join(HERE, 'emul_rt.cpp'),
              os.path.join(ROOT, 'include', 'hrfuser_hip.h'), os.path.join(ROOT, 'include', 'hrfuser_hip_debug.h')]
    for f in sorted(files):
        h.update(open(f, 'rb').read())
    h.update(b'asan' if sanitize else b'plain')
    h.update(b'group4')
    stamp = LIB + '.stamp'
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return LIB
    flags = ['-O1', '-g', '-fsanitize=address,undefined'] if sanitize else ['-O2']
    objs, procs = [], []
    for src in SOURCES + ['emul_rt.cpp']:
        sp = os.path.join(HERE if src.endswith('.cpp') else CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(OUT, src.replace('.hip', '.o').replace('.cpp', '.o'))
        cmd = ['g++', '-x', 'c++', '-std=c++17', '-fPIC', '-DHRF_EMUL', '-DHRF_GROUP_MAX=4', '-ffp-contract=off', f'-I{HERE}', f'-I{CSRC}',
               '-c', sp, '-o', obj] + flags
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'g++ failed on {src}')
    subprocess.check_call(['g++', '-shared', '-o', LIB] + objs + ['-lpthread'] + (['-fsanitize=address,undefined'] if sanitize else []))
    open(stamp, 'w').write(h.hexdigest())
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, sanitize='--asan' in sys.argv))
