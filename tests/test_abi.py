"""C-ABI boundary checks that need no GPU: the library loads, exports every symbol declared in
include/hrfuser_hip.h, the product path refuses CPU tensors (no fallback), registry/ctor contract."""
import copy
import ctypes
import json
import os

import pytest
import torch

from helpers import ROOT, load_cfgs
from hrfuser_amd import _lib


def test_header_symbols_exported():
    path = _lib.LIB_PATH
    assert os.path.exists(path), 'run `python -m hrfuser_amd.build_ext` (hipcc) first'
    import torch  # noqa: F401  (HIP runtime comes from the torch process)
    dll = ctypes.CDLL(path)
    protos = _lib.parse_header()
    assert len(protos) >= 19
    for name in protos:
        assert hasattr(dll, name), f'{name} declared in include/hrfuser_hip.h but not exported'
    assert hasattr(dll, 'hrf_build_digest')                     # (const char*: not a status-returning entry point)
    debug = _lib.parse_header(_lib.DEBUG_HEADER)
    assert set(debug) == {'hrf_debug_knob', 'hrf_wgrad_group_report', 'hrf_stamp', 'hrf_debug_spin'} and not set(debug) & set(protos)
    for name in debug:
        assert hasattr(dll, name), f'{name} declared in include/hrfuser_hip_debug.h but not exported'
    # ... and nothing ELSE is exported: every dynamic hrf_* symbol of the library is declared in one of the two headers
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith('hrf_')}
    allowed = set(protos) | set(debug) | {'hrf_build_digest', 'hrf_wgrad_stamps'}      # (-DHRF_WG_TIMING builds only)
    assert exported <= allowed, sorted(exported - allowed)


def test_library_contains_gfx950_code_object():
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob and b'amdgcn-amd-amdhsa' in blob


def test_no_cpu_fallback():
    from hrfuser_amd import build_backbone
    import helpers
    _lib.lib = helpers._REAL_LIB_FN          # the real HIP library, never the emulator
    net = build_backbone(copy.deepcopy(load_cfgs()['t_nus_bn']))
    x = torch.randn(1, 3, 32, 64)
    with pytest.raises(_lib.HRFuserHipError):
        net(x, [x.clone(), x.clone()])


def test_registry_and_errors():
    from hrfuser_amd import BACKBONES, HRFuserHRFormerBased, build_backbone
    assert BACKBONES.get('HRFuserHRFormerBased') is HRFuserHRFormerBased
    cfg = copy.deepcopy(load_cfgs()['t_nus'])
    net = build_backbone(cfg)
    with pytest.raises(Exception, match='num_fused_modalities does not fit'):
        net(torch.zeros(1, 3, 32, 64), [torch.zeros(1, 3, 32, 64)])
    bad = copy.deepcopy(load_cfgs()['t_nus'])
    bad['extra']['ModFusionA']['block'] = 'HRFORMER'
    with pytest.raises(Exception, match='Not valid fusion block'):
        build_backbone(bad)
    bad = copy.deepcopy(load_cfgs()['t_nus'])
    del bad['extra']['stage3']
    with pytest.raises(AssertionError):
        build_backbone(bad)
    assert net.train() is net and net.eval() is net


@pytest.mark.parametrize('tag', ['t_nus', 'b_nus', 't_stf'])
def test_state_dict_manifest(tag):
    from hrfuser_amd import build_backbone
    net = build_backbone(copy.deepcopy(load_cfgs()[tag]))
    with open(os.path.join(ROOT, 'tests', 'golden', f'state_manifest_{tag}.json')) as fh:
        man = json.load(fh)
    sd = net.state_dict()
    assert sum(p.numel() for p in net.parameters()) == man['n_params']
    assert [k for k in sd] and set(sd) == {e[0] for e in man['entries']}
    for k, shape, dt in man['entries']:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == 'torch.' + dt, k
    # optimizer paramwise rules of the reference key on these substrings
    assert any('relative_position_bias_table' in k for k in sd) and any('norm' in k for k in sd)


def test_pretrained_and_init_cfg(tmp_path):
    """hrnet.py:301-318: pretrained (str) == init_cfg Pretrained; both at once assert; non-str raises TypeError;
    zero_init_residual is dead code in the reference (hrnet.py:480-481) and must not change the initialisation."""
    from hrfuser_amd import build_backbone
    cfg = load_cfgs()['t_nus']
    torch.manual_seed(3)
    src = build_backbone(copy.deepcopy(cfg))
    with torch.no_grad():
        for p in src.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    path = str(tmp_path / 'ckpt.pth')
    torch.save({'state_dict': {'backbone.' + k: v for k, v in src.state_dict().items()}, 'meta': {}}, path)
    c2 = copy.deepcopy(cfg)
    c2['init_cfg'] = dict(type='Pretrained', checkpoint=path, prefix='backbone.')
    a = build_backbone(c2)
    for (k, p), (_, q) in zip(a.state_dict().items(), src.state_dict().items()):
        assert torch.equal(p, q), k
    plain = str(tmp_path / 'plain.pth')
    torch.save(src.state_dict(), plain)
    c3 = copy.deepcopy(cfg)
    c3['pretrained'] = plain
    with pytest.warns(UserWarning, match='pretrained is deprecated'):
        b = build_backbone(c3)
    assert all(torch.equal(p, q) for p, q in zip(b.state_dict().values(), src.state_dict().values()))
    c4 = copy.deepcopy(c3)
    c4['init_cfg'] = dict(type='Pretrained', checkpoint=plain)
    with pytest.raises(AssertionError):
        build_backbone(c4)
    c5 = copy.deepcopy(cfg)
    c5['pretrained'] = 5
    with pytest.raises(TypeError):
        build_backbone(c5)
    c6 = copy.deepcopy(cfg)
    c6['zero_init_residual'] = True
    z = build_backbone(c6)
    assert float(z.layer1[0].bn3.weight.min()) == 1.0          # norm3 NOT zeroed, as in the reference


def test_packaged_configs_equal_golden():
    """bench.py / smoke() take the resolved configs/hrfuser/*.py backbone dicts from the PACKAGE (hrfuser_amd/configs); they are
    the dicts the reference-importing generator wrote into tests/golden/ (oracle/tools/make_golden.py)"""
    import json
    from hrfuser_amd.configs import backbone_cfg
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'backbone_cfgs.json')))
    assert len(gold) >= 6
    for tag, cfg in gold.items():
        assert backbone_cfg(tag) == cfg, tag
    with pytest.raises(KeyError):
        backbone_cfg('no_such_config')
