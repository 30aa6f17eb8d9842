#!/usr/bin/env python
"""Every s_barrier must find the LDS writes of its own wave complete: hipcc (ROCm 7.2) once emitted the loop-header barrier
of a kernel whose bodies sit in a loop (round 4's persistent launch, docs/experiments/) WITHOUT the s_waitcnt lgkmcnt(0) in
front of it (a ds_write sat at the end of the previous iteration, behind the back edge), and two halves of a workgroup then
read different values.  The product has such loops too (host_kernel's chain workgroup, the K loops of every tile kernel).

Checks the gfx950 assembly of EVERY kernel: walking every basic block, a ds_write / ds_add that has not been followed by
an `s_waitcnt ... lgkmcnt(0)` must not reach an s_barrier -- within the block, or, for a block that ENDS with pending
writes, at the head of any block (conservatively: a block that starts with s_barrier before any lgkmcnt(0) wait is
reported when some block ends with pending LDS writes and can fall through or branch to it).

    python tools/check_barrier_waits.py            (compiles lcgp_amd/csrc/lcgp_hip.hip with -save-temps; CPU only)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'lcgp_amd', 'csrc', 'lcgp_hip.hip')


def kernel_bodies(asm, pattern=''):
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r'^(_Z\w+):', line)
        if m and pattern in m.group(1):
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.strip().startswith('.amdhsa_kernel') or line.startswith('.Lfunc_end'):
                out[name] = cur
                cur = None
                continue
            cur.append(line)
    return out


def check(lines):
    """returns (barriers, problems)"""
    blocks, label, cur = [], 'entry', []
    for ln in lines:
        s = ln.split(';')[0].strip()
        if not s:
            continue
        m = re.match(r'^(\.LBB\w+):', s)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        cur.append(s)
    blocks.append((label, cur))
    idx = {lab: i for i, (lab, _) in enumerate(blocks)}
    pend_out, head_barrier, succ = {}, {}, {}
    problems, nbar = [], 0
    for i, (lab, ins) in enumerate(blocks):
        pending, seen_wait, hb = False, False, False
        nxt = []
        falls = True
        for s in ins:
            op = s.split()[0]
            if op.startswith('ds_write') or op.startswith('ds_add') or op.startswith('ds_swizzle') is False and op.startswith('ds_store'):
                pending = True
            elif op == 's_waitcnt' and 'lgkmcnt(0)' in s:
                pending, seen_wait = False, True
            elif op == 's_barrier':
                nbar += 1
                if pending:
                    problems.append('%s: s_barrier with an LDS write of the same block pending' % lab)
                if not seen_wait:
                    hb = True
                seen_wait = True        # (a barrier orders nothing by itself, but what reaches the next one is a new question)
            elif op.startswith('s_cbranch') or op == 's_branch':
                t = s.split()[-1]
                nxt.append(t)
                if op == 's_branch':
                    falls = False
        if falls and i + 1 < len(blocks):
            nxt.append(blocks[i + 1][0])
        pend_out[lab], head_barrier[lab], succ[lab] = pending, hb, nxt
    # pending writes flowing into a block that reaches a barrier before any wait (one level of empty pass-through blocks)
    def reaches(lab, depth=0):
        if lab not in idx or depth > 6:
            return False
        if head_barrier[lab]:
            return True
        ins = blocks[idx[lab]][1]
        if any(s.split()[0] == 's_waitcnt' and 'lgkmcnt(0)' in s for s in ins):
            return False
        if any(s.split()[0].startswith('ds_') for s in ins):
            return False
        return any(reaches(t, depth + 1) for t in succ[lab])
    for lab, p in pend_out.items():
        if p:
            for t in succ[lab]:
                if reaches(t):
                    problems.append('%s ends with a pending LDS write and reaches the barrier at the head of %s' % (lab, t))
    return nbar, problems


def main():
    with tempfile.TemporaryDirectory() as td:
        res = subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-save-temps',
                              '-o', os.path.join(td, 'x.so'), SRC], cwd=td, capture_output=True, text=True)
        if res.returncode != 0:
            print(res.stderr)
            return 2
        path = [f for f in os.listdir(td) if f.endswith('gfx950.s')][0]
        asm = open(os.path.join(td, path)).read()
    rc = 0
    bodies = kernel_bodies(asm)
    assert bodies, 'no kernel found in the assembly'
    for name, lines in bodies.items():
        nbar, problems = check(lines)
        print('%s: %d barriers, %d problems' % (name, nbar, len(problems)))
        for p in problems[:20]:
            print('   ', p)
        rc |= bool(problems)
    return rc


if __name__ == '__main__':
    sys.exit(main())
