"""Per-kernel table of the EVAL forward (running-statistics BatchNorm) of a model: every C-ABI call of one eager forward recorded,
one representative per (entry point, shape) re-issued 20x in a captured hipGraph and timed with HIP events (as bench.py does for
the training step).   python tools/eval_kernels.py [t_nus_bn] [out.json]"""
import json
import os
import sys

import torch

os.environ.setdefault('HRF_LANES', '0')                       # record on one stream
os.environ['HRF_MODULE_GRAPH'] = '0'                          # eager launches: the module would replay a captured forward
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from hrfuser_amd import _lib, profiling                        # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 't_nus_bn'
args = bench.parse(['--model', tag])
dev = torch.device('cuda:0')
_, cfg, stf, H, W, mc, net, B, x, mods, cots, trainer = bench.build_workload(args, 0, 1, dev, None, False)
net.eval()
real = _lib.lib
base = real()
prof = profiling.ProfLib(base, timing=False)
with torch.no_grad():
    for _ in range(2):
        net(x, list(mods))
    torch.cuda.synchronize()
    _lib.lib = lambda: prof
    try:
        net(x, list(mods))
        torch.cuda.synchronize()
    finally:
        _lib.lib = real
sigs = {}
for name, a, cargs in prof.records:
    sg = profiling._signature(name, a)
    if sg in sigs:
        sigs[sg][0] += 1
    else:
        sigs[sg] = [1, name, a, cargs]
rows = []
for sg, (cnt, name, a, cargs) in sigs.items():
    fn = getattr(base, name)
    try:
        dt = profiling._graph_time(lambda fn=fn, cargs=cargs: fn(*cargs[:-1], _lib.stream_ptr()))
    except Exception as e:                                      # noqa: BLE001
        print('skip', name, e)
        continue
    rows.append(dict(shape=profiling.shape_tag(name, a), launches=cnt, us=dt * 1e6, ms=cnt * dt * 1e3))
rows.sort(key=lambda r: -r['ms'])
tot = sum(r['ms'] for r in rows)
print(f'{tag}: {len(prof.records)} launches per eval forward, {tot:.3f} ms isolated ({tot / B:.3f} ms / img)')
fam = {}
for r in rows:
    k = r['shape'].split('[')[0]
    f = fam.setdefault(k, [0, 0.0])
    f[0] += r['launches']; f[1] += r['ms']
for k, (n, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f'  {ms:7.3f} ms  n={n:4d}  {k}')
print()
for r in rows[:40]:
    print(f"{r['ms']:7.3f} ms  n={r['launches']:3d}  {r['us']:7.1f} us  {r['shape']}")
if len(sys.argv) > 2:
    json.dump(rows, open(sys.argv[2], 'w'), indent=1)
