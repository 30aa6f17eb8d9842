#!/usr/bin/env python
"""Per-kernel resource usage of the gfx950 code object (registers, spills, scratch, LDS, occupancy).

    python tools/codeobj_notes.py > profiles/rNN_codeobj_notes.txt

Compiles lcgp_amd/csrc/lcgp_hip.hip with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line
per kernel, de-mangled, sorted by name."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'lcgp_amd', 'csrc', 'lcgp_hip.hip')


def main():
    res = subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared',
                          '-Rpass-analysis=kernel-resource-usage', '-o', '/dev/null', SRC],
                         capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit(res.stderr)
    rows, cur = [], None
    for line in res.stderr.splitlines():
        m = re.search(r'remark:\s+(.*?)\s*\[-Rpass-analysis', line)
        if not m:
            continue
        key, _, val = m.group(1).partition(':')
        key, val = key.strip(), val.strip()
        if key == 'Function Name':
            cur = {'name': val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, nm in zip(rows, names):
        nm = nm.replace('(anonymous namespace)::', '')
        r['name'] = re.sub(r'\(.*$', '', nm).replace('void ', '')
    rows.sort(key=lambda r: r['name'])
    print('# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage lcgp_amd/csrc/lcgp_hip.hip')
    print('%-58s %5s %5s %5s %7s %7s %8s %6s' % ('kernel', 'VGPR', 'AGPR', 'SGPR', 'vspill', 'scratch', 'LDS', 'occ'))
    bad = 0
    for r in rows:
        print('%-58s %5s %5s %5s %7s %7s %8s %6s' % (r['name'][:58], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'),
                                                     r.get('VGPRs Spill'), r.get('ScratchSize [bytes/lane]'),
                                                     r.get('LDS Size [bytes/block]'), r.get('Occupancy [waves/SIMD]')))
        bad += int(r.get('VGPRs Spill', '0')) > 0
    print('# kernels with VGPR spills: %d' % bad)


if __name__ == '__main__':
    main()
