"""First-kernel offset per hardware queue in every stage of a tools/stage_trace_report.py listing (all stages shown):
    python tools/lane_starts.py gpurun_out/X/stage_trace.txt"""
import re
import sys

cur, first, order = None, {}, []
for line in open(sys.argv[1]):
    m = re.match(r'^(fwd|bwd|step) (\S+)\s+wall\s+([\d.]+) us\s+kernels\s+(\d+)', line)
    if m:
        cur = f'{m.group(1)} {m.group(2)}'
        order.append((cur, float(m.group(3)), int(m.group(4))))
        first[cur] = {}
        continue
    m = re.match(r'^\s+q\s*(\d+) s\s*\d+ \+\s*([\d.]+)\s+([\d.]+) us\s+(\S+)', line)
    if m and cur:
        q = m.group(1)
        if q not in first[cur]:
            first[cur][q] = (float(m.group(2)), m.group(4)[:22])
for name, wall, n in order:
    f = sorted(first[name].items(), key=lambda kv: kv[1][0])
    print(f'{name:26s} wall {wall:7.1f} n {n:4d}  ' + '  '.join(f'q{q}+{t:.0f}({k})' for q, (t, k) in f))
