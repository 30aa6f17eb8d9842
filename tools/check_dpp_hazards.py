"""Static check of the hand-written fp64 DPP instructions (inline asm in lcgp_hip.hip) against the hazard the compiler's
hazard recogniser cannot see inside an asm statement:

    a VALU instruction writes a VGPR  ->  a DPP instruction reads that VGPR as its DPP operand (src0):
    at least 2 wait states in between (s_nop N = N + 1 wait states; any other instruction = 1)

    python tools/check_dpp_hazards.py            # compiles the product source to gfx950 assembly, exits 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'lcgp_amd', 'csrc', 'lcgp_hip.hip')
REG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')


def regs(tok):
    m = REG.fullmatch(tok.strip().lstrip('-').strip('|'))
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def check(asm_text):
    """-> (number of DPP instructions, list of violations)"""
    bad, ndpp = [], 0
    window = []                      # (wait states this instruction provides, VGPRs it writes as a VALU op, text)
    for raw in asm_text.splitlines():
        line = raw.split(';')[0].strip()
        if not line or line.startswith(('.', '/')) or line.endswith(':'):
            if line.endswith(':'):
                window = []          # a label: control flow joins here, the compiler's own hazard handling ends the window
            continue
        op, _, rest = line.partition(' ')
        ops = [o.strip() for o in rest.split(',')] if rest else []
        if op.endswith('_dpp'):
            ndpp += 1
            src0 = regs(ops[1].split(' ')[0]) if len(ops) > 1 else set()
            ws = 0
            for w, wr, text in reversed(window):
                if ws >= 2:
                    break
                if wr & src0:
                    bad.append('%s   <- written %d wait state(s) earlier by: %s' % (line, ws, text))
                    break
                ws += w
        wait = int(ops[0]) + 1 if op == 's_nop' and ops and ops[0].isdigit() else 1
        written = regs(ops[0].split(' ')[0]) if op.startswith('v_') and ops and not op.startswith('v_cmp') else set()
        window.append((wait, written, line))
        window = window[-6:]
    return ndpp, bad


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        subprocess.check_call(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-o', out, SRC],
                              stderr=subprocess.DEVNULL)
        ndpp, bad = check(open(out).read())
    print('%d DPP instructions, %d hazard violations' % (ndpp, len(bad)))
    for b in bad[:20]:
        print('  ' + b)
    return 1 if bad or ndpp == 0 else 0


if __name__ == '__main__':
    sys.exit(main())
