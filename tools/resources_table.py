"""profiles/rNN_kernel_resources.txt from hrfuser_amd/kernel_resources.json (written by hrfuser_amd/build_ext.py from hipcc's
-Rpass-analysis=kernel-resource-usage remarks of the build that produced the shipped library): registers, scratch, static LDS
and the register-limited waves per SIMD of EVERY kernel, with the kernels that spill or sit at one wave per SIMD listed first,
each with its reason.      python tools/resources_table.py r05 > profiles/r05_kernel_resources.txt"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REASONS = [
    (r'attn_block_bwd_kernel<18, 1, 4, true, true, false>', 'one 8-byte value stored in the prologue and reloaded once in the epilogue (cross + FFN variant: 3 launches per step)'),
    (r'attn_block_fwd_kernel<144, 8>', '12x20 map: 12 windows for 256 CUs - one workgroup per CU whatever the registers; 9 channel tiles of accumulators per wave'),
    (r'wgrad3w_kernel', 'neck weight gradient (SURVEY 8f-1, not on the backbone step): 73.7 K accumulators per block in registers by design (36 MFMA tiles per wave); values spilled outside the pixel loop'),
    (r'lin_fwd_kernel<9, ', 'whole-row LayerNorm statistics of 144-channel rows at 12x20 (480 rows = 30 waves on the chip): occupancy is irrelevant, 9 channel tiles of accumulators per wave'),
]


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return re.sub(r'\(.*$', '', n)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'rNN'
    d = json.load(open(os.path.join(ROOT, 'hrfuser_amd', 'kernel_resources.json')))
    rows = sorted(((short(k), v) for k, v in d['kernels'].items()), key=lambda kv: (kv[1]['file'], kv[0]))
    print('# Registers / scratch / static LDS / waves per SIMD of EVERY kernel of the shipped library (hipcc -Rpass-analysis=kernel-resource-usage of the')
    print(f'# build that produced hrfuser_amd/libhrfuser_hip.so; written by hrfuser_amd/build_ext.py into kernel_resources.json, digest {d["digest"][:16]}; {tag}).')
    print('# waves/SIMD is the REGISTER-limited occupancy (dynamic LDS can lower it: attn_block_bwd<36,...> 128 KB -> one 8-wave workgroup per CU).')
    print('\n## exceptions (scratch > 0 or 1 wave per SIMD), with the reason')
    n_exc = 0
    for name, v in rows:
        if v['scratch'] > 0 or v['waves_per_simd'] <= 1:
            why = next((w for pat, w in REASONS if name.startswith(pat)), 'NO REASON ON FILE')
            print(f'{name:70s} vgpr {v["vgpr"]} agpr {v["agpr"]} scratch {v["scratch"]} B/lane  waves/SIMD {v["waves_per_simd"]}  -- {why}')
            n_exc += 1
    if not n_exc:
        print('(none)')
    print(f'\n## all kernels ({len(rows)})')
    print(f'{"kernel":86s} {"VGPR":>5s} {"AGPR":>5s} {"scratch":>8s} {"LDS(stat)":>10s} {"w/SIMD":>6s}  file')
    for name, v in rows:
        print(f'{name:86s} {v["vgpr"]:5d} {v["agpr"]:5d} {v["scratch"]:8d} {v["lds_static"]:10d} {v["waves_per_simd"]:6d}  {v["file"]}')


if __name__ == '__main__':
    main()
