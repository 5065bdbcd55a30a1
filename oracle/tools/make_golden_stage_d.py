"""Pre-neck fusion (LidarStageD / ModFusionD, hrfuser_hrformer_based.py:454-468,609-625) is disabled in every reference
config; this container-only script enables it on the HRFuser-T nus config (stage D = a copy of LidarStageC with one module,
ModFusionD = a copy of ModFusionC), checks oracle.HRFuserOracle bit-exact against the reference class (eval + train outputs,
input gradient) and writes the resolved config to tests/golden/backbone_cfg_stage_d.json."""
import sys, copy, torch
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/oracle/tools')
import hrfuser_oracle as O, ref_loader as R
cfg = R.backbone_cfg('cascade_rcnn_hrfuser_t_1x_nus_r640_l_r_fusion_bn')
ex = cfg['extra']
ex['LidarStageD'] = copy.deepcopy(ex['LidarStageC']); ex['LidarStageD']['num_modules'] = 1
ex['ModFusionD'] = copy.deepcopy(ex['ModFusionC'])
print({k: ex['ModFusionD'][k] for k in ('num_branches','num_channels','block')}, ex['LidarStageD'])
ref = R.build_reference(copy.deepcopy(cfg))
c2 = copy.deepcopy(cfg); c2.pop('type')
orc = O.HRFuserOracle(**c2)
assert list(ref.state_dict().keys()) == list(orc.state_dict().keys()), set(ref.state_dict()) ^ set(orc.state_dict())
O.seeded_fill_(ref, 0); O.seeded_fill_(orc, 0)
from make_golden import disable_stochastic
disable_stochastic(ref); disable_stochastic(orc)
x, mods = O.seeded_inputs(2, 64, 96, [3,3], seed=1)
for mode in (False, True):
    ref.train(mode); orc.train(mode)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = ref(xa, [m.clone() for m in mods]); yb = orc(xb, [m.clone() for m in mods])
    for a,b in zip(ya,yb): assert float((a-b).abs().max())==0.0
    sum(t.sum() for t in ya).backward(); sum(t.sum() for t in yb).backward()
    assert float((xa.grad-xb.grad).abs().max())==0.0
    print('mode', mode, 'bit-exact', [tuple(t.shape) for t in ya])
import json
json.dump({'t_nus_bn_stage_d': json.loads(json.dumps(cfg, default=lambda o: list(o)))}, open('/root/repo/tests/golden/backbone_cfg_stage_d.json','w'), indent=1, sort_keys=True)
