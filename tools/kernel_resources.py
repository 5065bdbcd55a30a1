#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage,
device-only compile with the library's flags).  Usage: python tools/kernel_resources.py attn_block.hip [extra flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrfuser_amd import build_ext  # noqa: E402


def main():
    src = sys.argv[1]
    if not os.path.exists(src):
        src = os.path.join(build_ext.CSRC, src)
    cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '-x', 'hip', f'--offload-arch={build_ext.ARCH}'] + build_ext.FLAGS + \
        sys.argv[2:] + ['--offload-device-only', '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r'remark:\s+([A-Za-z][A-Za-z \[\]/]*?):\s+(\S+) \[-Rpass', line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == 'Function Name':
            cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '')[:90]}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    print(f'{"kernel":92s} VGPR AGPR scratch  LDS(static) occ')
    for r in rows:
        print(f'{r["name"]:92s} {r.get("VGPRs", "?"):>4s} {r.get("AGPRs", "?"):>4s} {r.get("ScratchSize [bytes/lane]", "?"):>7s} '
              f'{r.get("LDS Size [bytes/block]", "?"):>12s} {r.get("Occupancy [waves/SIMD]", "?"):>3s}  spill {r.get("VGPRs Spill", "?")}')


if __name__ == '__main__':
    main()
