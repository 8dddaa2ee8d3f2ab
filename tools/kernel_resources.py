#!/usr/bin/env python3
"""Registers / scratch / occupancy / LDS of every kernel in one .hip file, from hipcc's -Rpass-analysis=kernel-resource-usage
(no GPU needed).  Usage: python tools/kernel_resources.py gnndelete_amd/csrc/spmm.hip [name-substring]"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', f'-I{ROOT}/include', '-c', src, '-o',
           '/dev/null', '-Rpass-analysis=kernel-resource-usage']
    if src.endswith('.cpp'):
        cmd[1:1] = ['-x', 'hip']
    err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r'remark: (?:\s*)Function Name: (\S+)', line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    print('| kernel | VGPRs | AGPRs | scratch B/lane | waves/SIMD | LDS B |')
    print('|---|---|---|---|---|---|')
    for name, r in rows.items():
        dem = subprocess.run(['c++filt', name], stdout=subprocess.PIPE, text=True).stdout.strip()
        short = dem.split('(')[0].replace('void ', '')
        if flt and flt not in short:
            continue
        print(f"| `{short}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('ScratchSize')} | {r.get('Occupancy')} | {r.get('LDS Size')} |")


if __name__ == '__main__':
    main()
