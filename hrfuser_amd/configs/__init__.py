"""Resolved `model.backbone` dicts of the reference's configs/hrfuser/*.py (BASELINE.json configs): what mmcv's Config produces
for cascade_rcnn_hrfuser_{t,b}_1x_nus_r640_l_r_fusion.py and ..._t_1x_stf_r1248_4mod.py, with the SyncBN (`t_nus`) and
single-GPU BN (`t_nus_bn`) norm_cfg variants.  Data only; bench.py and __graft_entry__.smoke() build their workloads from
here (tests/golden/backbone_cfgs.json is the same file as the reference-importing generator wrote it -
tests/test_abi.py::test_packaged_configs_equal_golden keeps the two identical)."""
import copy
import json
import os

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'backbone_cfgs.json')
_CACHE = None


def backbone_cfg(tag):
    """-> a fresh copy of the backbone dict of `tag` (t_nus[_bn] | b_nus[_bn] | t_stf[_bn])"""
    global _CACHE
    if _CACHE is None:
        with open(_PATH) as fh:
            _CACHE = json.load(fh)
    if tag not in _CACHE:
        raise KeyError(f'{tag!r}: known configurations {sorted(_CACHE)}')
    return copy.deepcopy(_CACHE[tag])
