"""python tools/kern_grep.py kernels.json PATTERN... -> the per-signature rows of a bench.py --dump-kernels file whose shape / kernel matches"""
import json
import re
import sys
d = json.load(open(sys.argv[1]))
pats = sys.argv[2:] or ['.']
for s in d['signatures']:
    txt = s['shape'] + ' ' + s['kernel']
    if any(re.search(p, txt) for p in pats):
        fl = s['flops_per_launch'] / s['avg_launch_us'] / 1e6
        bw = s['bytes_per_launch'] / s['avg_launch_us'] / 1e6
        print(f"{s['time_per_step_ms']:.3f} n={s['launches_per_step']:3d} {s['avg_launch_us']:7.1f}us {fl:6.1f}TF {bw:5.2f}TB/s  {s['kernel'][:34]:34s} {s['shape']}")
print('launches per step', d.get('launches_per_step'))
