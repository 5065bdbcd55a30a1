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
RES = os.path.join(HERE, 'kernel_resources.json')     # registers / scratch / LDS / waves per SIMD of every kernel of the build
SOURCES = ['conv_engine.hip', 'wgrad_tiled.hip', 'lin_engine.hip', 'lin2_engine.hip', 'conv3_engine.hip', 'conv3x_engine.hip', 'wgrad3x_engine.hip', 'conv3w_engine.hip', 'dwconv.hip', 'attention.hip', 'attn_block.hip', 'ffn_eval.hip', 'pointwise.hip', 'group.hip', 'p2p_exchange.hip']
HEADERS = ['hrf_rt.h', 'hrf_common.h', 'hrf_lin.h', 'hrf_group.h', 'hrf_wgrad.h', os.path.join(ROOT, 'include', 'hrfuser_hip.h'),
           os.path.join(ROOT, 'include', 'hrfuser_hip_debug.h')]
ARCH = 'gfx950'


FLAGS = ['-O3', '-std=c++17', '-fPIC', '-ffp-contract=off'] + os.environ.get('HRF_EXTRA_FLAGS', '').split()   # e.g. -DHRF_AB_TIMING


def _digest():
    """sha256 over every source, header, the compile flags and the target arch."""
    h = hashlib.sha256()
    h.update((ARCH + ' ' + ' '.join(FLAGS)).encode())
    for f in SOURCES + HEADERS:
        p = f if os.path.isabs(f) else os.path.join(CSRC, f)
        if os.path.exists(p):
            with open(p, 'rb') as fh:
                h.update(f.encode() + b'\0' + fh.read())
    return h.hexdigest()


def embedded_digest(path=LIB):
    """The source digest compiled INTO the library (symbol hrf_build_digest), read from the file's bytes so that a
    stale binary is never dlopen'ed just to find out that it is stale."""
    try:
        blob = open(path, 'rb').read()
    except OSError:
        return None
    tag = b'HRF_BUILD_DIGEST='
    i = blob.find(tag)
    return blob[i + len(tag):i + len(tag) + 64].decode('ascii', 'replace') if i >= 0 else None


def _torch_lib_dir():
    try:
        import torch
        return os.path.join(os.path.dirname(torch.__file__), 'lib')
    except Exception:
        return None


def compile_one(src, obj, verbose=True, remarks=None):
    """remarks: a file that receives the compiler's kernel-resource-usage remarks (stderr) - no influence on the code."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '-x', 'hip', f'--offload-arch={ARCH}'] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    if remarks is None:
        return subprocess.Popen(cmd)
    return subprocess.Popen(cmd + ['-Rpass-analysis=kernel-resource-usage'], stderr=open(remarks, 'w'))


def parse_remarks(text):
    """hipcc -Rpass-analysis=kernel-resource-usage output -> [{name, vgpr, agpr, sgpr, scratch, lds, waves_per_simd, spill}]"""
    import re
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r'remark:\s+([A-Za-z][A-Za-z \[\]/]*?):\s+(\S+) \[-Rpass', line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == 'Function Name':
            cur = {'mangled': v}
            rows.append(cur)
        elif cur is not None:
            key = {'VGPRs': 'vgpr', 'AGPRs': 'agpr', 'SGPRs': 'sgpr', 'ScratchSize [bytes/lane]': 'scratch',
                   'LDS Size [bytes/block]': 'lds_static', 'Occupancy [waves/SIMD]': 'waves_per_simd', 'VGPRs Spill': 'spill'}.get(k)
            if key:
                cur[key] = int(v)
    if rows:
        try:                                       # (binutils may be absent: the table is a measurement aid - keep mangled names then)
            names = subprocess.run(['c++filt'], input='\n'.join(r['mangled'] for r in rows), capture_output=True, text=True).stdout.split('\n')
        except OSError:
            names = []
        if len(names) < len(rows):
            names = [r['mangled'] for r in rows]
        for r, n in zip(rows, names):
            r['name'] = n.strip().replace('(anonymous namespace)::', '')
            r['name'] = r['name'][5:] if r['name'].startswith('void ') else r['name']
            r['name'] = r['name'].split('(')[0]
            del r['mangled']
    return rows


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the C-ABI shared library.  The library is rebuilt whenever the
    digest embedded in it differs from the digest of the sources in the tree (no side-car stamp file: a pulled
    source change can never be paired with an old binary)."""
    dg = _digest()
    if not force and embedded_digest() == dg:
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        procs.append((src, compile_one(src, obj, verbose, remarks=obj + '.remarks')))
        objs.append(obj)
    dsrc = os.path.join(CSRC, '_build_digest.gen.cpp')
    with open(dsrc, 'w') as fh:
        fh.write('// generated by hrfuser_amd/build_ext.py - do not edit\n'
                 f'extern "C" const char* hrf_build_digest(void) {{ return "HRF_BUILD_DIGEST={dg}" + 17; }}\n')
    dobj = dsrc.replace('.cpp', '.o')
    subprocess.check_call(['g++', '-O1', '-fPIC', '-c', dsrc, '-o', dobj])
    objs.append(dobj)
    for src, p in procs:
        if p.wait() != 0:
            sys.stderr.write(open(os.path.join(CSRC, src.replace('.hip', '.o.remarks'))).read()[-4000:])
            raise RuntimeError(f'hipcc failed on {src}')
    # the resource table of THIS build (read by bench.py's roofline and tools/kernel_resources.py; carries the source digest)
    import json
    table = {}
    for src, _ in procs:
        rp = os.path.join(CSRC, src.replace('.hip', '.o.remarks'))
        try:
            text = open(rp).read()
            # the compiler's stderr went into the remarks file: everything in it that is NOT a resource remark (warnings of a
            # successful compile) is shown, as it would have been without the table
            lines = text.splitlines()
            other = [k for k, ln in enumerate(lines) if (' warning: ' in ln or ' error: ' in ln)]
            if other and verbose:
                sys.stderr.write('\n'.join('\n'.join(lines[k:k + 3]) for k in other[-20:]) + '\n')
            for r in parse_remarks(text):
                r['file'] = src
                table[r.pop('name')] = r
            os.remove(rp)
        except Exception as e:                     # never fail a build over its resource table
            sys.stderr.write(f'build_ext: resource remarks of {src} not parsed ({type(e).__name__}: {e})\n')
    try:
        with open(RES, 'w') as fh:
            json.dump({'digest': dg, 'arch': ARCH, 'flags': FLAGS, 'kernels': table}, fh, indent=0, sort_keys=True)
    except OSError as e:
        sys.stderr.write(f'build_ext: {RES} not written ({e})\n')
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
    assert embedded_digest() == dg
    return LIB


def resources():
    """{kernel name: {vgpr, agpr, scratch, lds_static, waves_per_simd, spill, file}} of the shipped build, or {} when the
    table on disk belongs to other sources."""
    import json
    try:
        with open(RES) as fh:
            j = json.load(fh)
    except (OSError, ValueError):
        return {}
    return j.get('kernels', {}) if j.get('digest') == embedded_digest() else {}


def compile_check(src='attention.hip'):
    """Cross-compile ONE translation unit for gfx950 into a scratch object: proves the toolchain works on this machine
    even when the shipped library is current (used by __graft_entry__.build)."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        obj = os.path.join(td, 'check.o')
        if compile_one(src, obj, verbose=False).wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
        blob = open(obj, 'rb').read()
        assert ARCH.encode() in blob, 'no gfx950 code object in the compiled unit'
    return True


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
