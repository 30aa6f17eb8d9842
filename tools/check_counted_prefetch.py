#!/usr/bin/env python
"""The hand-counted operand prefetch of the tile kernels (lcgp_hip.hip: gload_piece / vm_wait_set) issues its global loads
through inline asm, which hipcc's wait-count pass does not track: the data is only valid behind the explicit
`s_waitcnt vmcnt(N)` the source places in front of the first use.  That holds as long as the compiler never touches an
asm-loaded register between the load and the wait -- e.g. a move at a control-flow join (seen once while the prefetch was
being written: a copy placed right behind the load copied the register before the data had arrived), or a spill.

Checks the gfx950 assembly of every kernel that contains such loads:
  * between an asm load of v[a:b] and the next asm `s_waitcnt vmcnt`, no instruction reads or writes v[a:b];
  * the kernel has no scratch (spill) traffic inside a loop that contains asm loads -- a spill reload is a vector-memory
    load: the compiler waits for it with vmcnt(0), i.e. for every prefetched stage.

    python tools/check_counted_prefetch.py          (compiles lcgp_amd/csrc/lcgp_hip.hip with -save-temps; CPU only)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'lcgp_amd', 'csrc', 'lcgp_hip.hip')
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from check_barrier_waits import kernel_bodies  # noqa: E402

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(lines):
    """returns (number of asm loads, problems)"""
    problems, nload = [], 0
    pending = {}            # register -> line number of the asm load that is in flight into it
    in_asm = False
    loop_depth_has_asm = False
    loop_start = None
    loops = []              # (first line, last line) of inner loops, by their header comment and back branch
    for i, ln in enumerate(lines):
        s = ln.split(';')[0].strip() if not ln.strip().startswith(';;#') else ln.strip()
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not s or s.endswith(':') and not s.startswith('v_'):
            continue
        if in_asm:
            if s.startswith('global_load_dwordx4'):
                dst = regs_of(s.split(',')[0])
                nload += 1
                for r in dst:
                    pending[r] = i
            elif s.startswith('s_waitcnt') and 'vmcnt' in s:
                n = int(re.search(r'vmcnt\((\d+)\)', s).group(1))
                # in-order return: everything but the n youngest loads is complete
                order = sorted(set(pending.values()))
                keep = set(order[len(order) - n:]) if n > 0 else set()
                pending = {r: at for r, at in pending.items() if at in keep}
            continue
        if pending:
            touched = regs_of(s) & set(pending)
            if touched:
                problems.append('line %d: `%s` touches v%s while the asm load of line %d is in flight'
                                % (i, s, sorted(touched)[:4], pending[sorted(touched)[0]]))
    # spill traffic inside loops that hold asm loads
    header = None
    for i, ln in enumerate(lines):
        if 'Inner Loop Header' in ln:
            header = i
        m = re.match(r'^\s*s_cbranch_\w+\s+(\.LBB\w+)', ln)
        if m and header is not None:
            # a back branch closes the innermost open loop if its target label lies at or before the header
            lab = m.group(1) + ':'
            tgt = next((j for j in range(header - 3, header + 1) if j >= 0 and lines[j].strip().startswith(lab)), None)
            if tgt is not None:
                loops.append((tgt, i))
                header = None
    for a, b in loops:
        body = lines[a:b + 1]
        if any('global_load_dwordx4' in x and k > 0 and ';;#ASMSTART' in body[k - 1] for k, x in enumerate(body)):
            for k, x in enumerate(body):
                if 'scratch_' in x:
                    problems.append('line %d: spill traffic `%s` inside a loop with hand-counted loads' % (a + k, x.strip()))
    return nload, problems


def main():
    with tempfile.TemporaryDirectory() as td:
        res = subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-save-temps',
                              '-o', os.path.join(td, 'x.so'), SRC], cwd=td, capture_output=True, text=True)
        if res.returncode != 0:
            print(res.stderr)
            return 2
        path = [f for f in os.listdir(td) if f.endswith('gfx950.s')][0]
        asm = open(os.path.join(td, path)).read()
    rc, nk, total = 0, 0, 0
    for name, lines in kernel_bodies(asm).items():
        n, problems = check(lines)
        if n:
            nk += 1
            total += n
        for pr in problems:
            print('%s: %s' % (name, pr))
            rc = 1
    print('%d kernels with hand-counted loads, %d asm loads, %s' % (nk, total, 'no findings' if rc == 0 else 'FINDINGS'))
    assert nk > 0, 'no kernel with hand-counted loads found: has the prefetch been renamed?'
    return rc


if __name__ == '__main__':
    sys.exit(main())
