"""ctypes binding of libhrfuser_hip.so — the C-ABI boundary (include/hrfuser_hip.h).

Prototypes are derived from the public header itself, so the header is the single source of
truth for the ABI.  There is NO CPU fallback: if the HIP library is missing or a tensor is not
on a ROCm device the call raises.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'hrfuser_hip.h')
DEBUG_HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'hrfuser_hip_debug.h')      # measurement / tuning entry points
# HRF_LIB_PATH: another build of the same library (same-box A/B of two kernel versions: tools/gpu_ab_lib.sh)
LIB_PATH = os.environ.get('HRF_LIB_PATH') or os.path.join(_HERE, 'libhrfuser_hip.so')

def _header_int(name, default):
    m = re.search(r'#define\s+' + name + r'\s+(\d+)', open(HEADER).read())
    return int(m.group(1)) if m else default


STAT_COPIES = _header_int('HRF_STAT_COPIES', 16)   # replication of cross-block accumulators (see header)
FIN_MAXC = _header_int('HRF_FIN_MAXC', 576)        # widest BatchNorm the consumer kernels finalise on load

def struct_from_header(tag, path=HEADER):
    """ctypes mirror of `typedef struct <tag> {...} <tag>_t;` in the public header (the header is the single source of
    truth for field order and types: pointers -> c_void_p, int / long / float / double by value)."""
    src = re.sub(r'/\*.*?\*/', '', open(path).read(), flags=re.S)
    m = re.search(r'typedef\s+struct\s+' + tag + r'\s*\{(.*?)\}\s*' + tag + r'_t\s*;', src, flags=re.S)
    if not m:
        raise KeyError(tag)
    fields = []
    for decl in m.group(1).split(';'):
        decl = ' '.join(decl.split())
        if not decl:
            continue
        arr = re.search(r'(\w+)\[(\w+)\]$', decl)
        if arr:                                        # fixed-size array field (extent: a number or a #define of the header)
            ext = int(arr.group(2)) if arr.group(2).isdigit() else _header_int(arr.group(2), 0)
            base = decl.split(' ', 1)[0]
            el = ctypes.c_void_p if '*' in decl else {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'double': ctypes.c_double}[base]
            fields.append((arr.group(1), el * ext))
            continue
        if '*' in decl:
            ct, names = ctypes.c_void_p, [re.findall(r'(\w+)$', decl)[0]]
        else:
            base, rest = decl.split(' ', 1)
            ct = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'double': ctypes.c_double}[base]
            names = [n.strip() for n in rest.split(',')]
        fields += [(n, ct) for n in names]
    return type(tag, (ctypes.Structure,), {'_fields_': fields})


BnFin = struct_from_header('hrf_bn_fin')            # BatchNorm of a consumer's input finalised on load
BnBFin = struct_from_header('hrf_bn_bfin')          # BatchNorm-backward coefficients derived on load
AttnBlock = struct_from_header('hrf_attn_block')    # fused window-attention block (csrc/attn_block.hip)
FfnEval = struct_from_header('hrf_ffn_eval')        # eval-mode CrossFFN in one launch (csrc/ffn_eval.hip)
Conv3xPackJob = struct_from_header('hrf_conv3x_pack_job')   # one weight tensor of hrf_conv3x_pack (csrc/conv3x_engine.hip)
P2p = struct_from_header('hrf_p2p')                 # peer-to-peer SyncBN exchange context (csrc/p2p_exchange.hip)


def _ptr(t):
    return None if t is None else t.data_ptr()


_RAW_RETURN = ('hrf_conv3_wgrad_wide_scratch', 'hrf_conv_bwd_weight_scratch', 'hrf_conv3x_pack_size', 'hrf_conv3x_supported', 'hrf_conv_fwd_split_scratch', 'hrf_wgrad_group_report', 'hrf_attn_block_supported', 'hrf_attn_block_bwd_supported', 'hrf_group_count',
               'hrf_ffn_eval_supported')       # return a value, not a status
_ERR = {1: 'HRF_ERR_ARG (bad argument)', 2: 'HRF_ERR_LAUNCH (kernel launch failed)'}


class HRFuserHipError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: [(ctype, argname), ...]} for every `int hrf_*(...)` declared in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(?:int|long)\s+(hrf_\w+)\s*\(([^)]*)\)\s*;', src):
        args = []
        for a in m.group(2).split(','):
            a = ' '.join(a.split())
            if not a or a == 'void':
                continue
            name = re.findall(r'(\w+)$', a)[0]
            if '*' in a:
                ct = ctypes.c_void_p
            elif a.startswith('float'):
                ct = ctypes.c_float
            elif a.startswith('double'):
                ct = ctypes.c_double
            elif a.startswith('long'):
                ct = ctypes.c_long
            else:
                ct = ctypes.c_int
            args.append((ct, name))
        protos[m.group(1)] = args
    return protos


class Lib:
    """Loaded C-ABI library; attribute access returns checked callables taking tensors/ints."""

    def __init__(self, path, require_cuda=True):
        if not os.path.exists(path):
            raise HRFuserHipError(
                f'{path} not found: the HRFuser HIP extension is not built. Run '
                '`python -m hrfuser_amd.build_ext` (needs hipcc). There is no CPU fallback.')
        self.path = path
        self.require_cuda = require_cuda
        self._dll = ctypes.CDLL(path)
        self.protos = parse_header()
        self.protos.update(parse_header(DEBUG_HEADER))     # (bound for bench.py / tools / tests; the product path calls none of them)
        self._fns = {}
        for name, args in self.protos.items():
            fn = getattr(self._dll, name)          # AttributeError if a declared symbol is missing
            fn.restype = ctypes.c_long if name in ('hrf_conv3_wgrad_wide_scratch', 'hrf_conv_bwd_weight_scratch', 'hrf_conv3x_pack_size', 'hrf_conv_fwd_split_scratch', 'hrf_wgrad_group_report', 'hrf_group_count') else ctypes.c_int
            fn.argtypes = [ct for ct, _ in args]
            self._fns[name] = self._wrap(name, fn, args)

    def _wrap(self, name, fn, args):
        kinds = [ct for ct, _ in args]
        nargs = len(kinds)
        require_cuda = self.require_cuda

        def call(*a):
            if len(a) != nargs:
                raise TypeError(f'{name} expects {nargs} args, got {len(a)}')
            conv = []
            for v, ct in zip(a, kinds):
                if ct is ctypes.c_void_p:
                    if v is None:
                        conv.append(None)
                    elif isinstance(v, (ctypes.Structure, ctypes.Array)):
                        conv.append(ctypes.addressof(v))
                    elif isinstance(v, torch.Tensor):
                        if require_cuda and not v.is_cuda:
                            raise HRFuserHipError(f'{name}: tensor on {v.device}; the HIP path has no CPU fallback')
                        conv.append(v.data_ptr())
                    else:
                        conv.append(int(v))
                else:
                    conv.append(v)
            rc = fn(*conv)
            if name in _RAW_RETURN:
                return rc
            if rc != 0:
                raise HRFuserHipError(f'{name} failed: {_ERR.get(rc, rc)}')
        call.__name__ = name
        return call

    def __getattr__(self, name):
        try:
            return self.__dict__['_fns'][name]
        except KeyError:
            raise AttributeError(name)


_LIB = None


def lib():
    """The process-wide HIP library handle (loaded lazily, after torch so its HIP runtime is shared)."""
    global _LIB
    if _LIB is None:
        _LIB = Lib(LIB_PATH, require_cuda=True)
        for kv in os.environ.get('HRF_KNOBS', '').split(','):      # tuning experiments: "key=value,..."
            if '=' in kv:
                k, v = kv.split('=')
                _LIB.hrf_debug_knob(int(k), int(v))
    return _LIB


def stream_ptr():
    """hipStream_t of torch's current stream (0 when no GPU context: emulation tests only)."""
    if torch.cuda.is_available():
        return torch.cuda.current_stream().cuda_stream
    return 0
