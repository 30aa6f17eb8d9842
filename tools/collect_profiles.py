"""Copies the artefacts of one tools/profile_round.sh run (gpurun_out/<tag>/) into profiles/ under the round's names and
re-derives profiles/traffic.json and the per-kernel summaries from the counter passes.

    python tools/collect_profiles.py <tag> <round prefix, e.g. r02>
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def summary(d):
    return subprocess.check_output([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), d]).decode()


def per_kernel(d, counter):
    cc = list(csv.DictReader(open(max(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime))))
    out = {}
    for r in cc:
        if r['Counter_Name'] != counter:
            continue
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
        e = out.setdefault(n, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += float(r['Counter_Value'])
        e[2] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e6
    return out


def main():
    tag, pre = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles')
    for a, b in (('bench.json', 'bench.json'), ('bench_cfg2.json', 'bench_cfg2.json'), ('bench_cfg4.json', 'bench_cfg4.json'),
                 ('bench_cfg5.json', 'bench_cfg5.json'), ('bench_q4.json', 'bench_q4.json'), ('bench_q2.json', 'bench_q2.json'),
                 ('bench_q1.json', 'bench_q1.json'), ('bench_cfg5_q2.json', 'bench_cfg5_q2.json'), ('bench_cfg5_q1.json', 'bench_cfg5_q1.json'),
                 ('bench_gloo2.json', 'bench_gloo2_one_gpu.json'),
                 ('host_overhead.txt', 'host_overhead.txt'), ('timeline_q1_progressive.txt', 'timeline_q1_progressive.txt'),
                 ('timeline_q1_classic.txt', 'timeline_q1_classic.txt'), ('fit_wallclock.txt', 'fit_wallclock.txt'),
                 ('timeline_q8.txt', 'timeline_q8.txt')):
        if not os.path.exists(os.path.join(src, a)):
            continue
        shutil.copy(os.path.join(src, a), os.path.join(dst, '%s_%s' % (pre, b)))
    shutil.copy(max(glob.glob(src + '/stats/*/*kernel_stats.csv'), key=os.path.getmtime), os.path.join(dst, pre + '_bench_kernel_stats.csv'))
    for d, name in (('pmc_fetch', 'pmc_fetch_size'), ('pmc_write', 'pmc_write_size'), ('pmc_valu', 'pmc_valu_cfg3'),
                    ('pmc_valu_cfg4', 'pmc_valu_cfg4'), ('pmc_mfma', 'pmc_mfma'), ('pmc_mfma_q1', 'pmc_mfma_q1')):
        if not os.path.isdir(os.path.join(src, d)):
            continue
        open(os.path.join(dst, '%s_%s.txt' % (pre, name)), 'w').write(summary(os.path.join(src, d)))
    bench = json.loads([ln for ln in open(os.path.join(src, 'bench.json')).read().strip().splitlines() if ln.startswith('{')][-1])
    lib_hash = subprocess.check_output([sys.executable, '-c', 'import sys; sys.path.insert(0, %r); from lcgp_amd import _hip; '
                                        'print(_hip.source_hash())' % ROOT]).decode().strip()
    fetch = per_kernel(os.path.join(src, 'pmc_fetch'), 'FETCH_SIZE')
    write = per_kernel(os.path.join(src, 'pmc_write'), 'WRITE_SIZE')
    lau = [k for k in fetch if 'tile_gemm<double, 4, 128, 8>' in k][0]
    f_kb = fetch[lau][1] / fetch[lau][0]
    w_kb = write[lau][1] / write[lau][0]
    traffic = {
        'lib_hash': lib_hash,
        'tile_gemm_lauum_bytes_per_launch': (2.0 * f_kb + w_kb) * 1024.0,
        'how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only '
               '(tools/profile_round.sh; summaries in profiles/%s_pmc_fetch_size.txt / %s_pmc_write_size.txt), 3 launches each, '
               'n=4096 q=8 fp64; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads; '
               'cross-check in the same pass: grad_kernel reads the 545 MB of lower tiles once and reports half), WRITE_SIZE '
               'exact; units KB. lib_hash = lcgp_source_hash() of the library the counters were collected on: bench.py only '
               'reports this number when the loaded library has the same hash' % (pre, pre),
        'fetch_kb_per_launch_raw': f_kb,
        'write_kb_per_launch': w_kb,
        'algorithmic_bytes_per_launch': 1073741824.0,
    }
    try:
        mf = per_kernel(os.path.join(src, 'pmc_mfma'), 'SQ_VALU_MFMA_BUSY_CYCLES')
        ga = per_kernel(os.path.join(src, 'pmc_mfma'), 'GRBM_GUI_ACTIVE')
        traffic['tile_gemm_lauum_mfma_busy'] = mf[lau][1] / (1024.0 * ga[lau][1] / 8.0)
    except Exception as exc:             # (no MFMA pass in this run)
        print('no MFMA-busy figure:', exc)
    json.dump(traffic, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
    # HBM traffic and MFMA busy per kernel and evaluation (3 evaluations per pass)
    lines = ['# HBM-side traffic per evaluation and kernel (n=4096 q=8 fp64), from %s_pmc_fetch_size.txt / %s_pmc_write_size.txt' % (pre, pre),
             '# (3 evaluations each; FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as is; durations are those of the profiled runs)']
    rows = []
    for k in fetch:
        gb = (2.0 * fetch[k][1] + write.get(k, [0, 0.0, 0.0])[1]) * 1024.0 / 3e9
        ms = fetch[k][2] / 3.0
        rows.append((gb, '%-46s launches/eval %5.1f %7.3f ms/eval %6.2f GB/eval %5.2f TB/s' % (k[5:51] if k.startswith('void ') else k[:46], fetch[k][0] / 3.0, ms, gb, gb / ms if ms else 0.0)))
    lines += [r[1] for r in sorted(rows, key=lambda r: -r[0])[:12]]
    open(os.path.join(dst, pre + '_hbm_traffic_per_kernel.txt'), 'w').write('\n'.join(lines) + '\n')
    lines = ['# MFMA pipe busy per kernel = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8)   (256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs)',
             '# from %s_pmc_mfma.txt (3 evaluations, n=4096 q=8 fp64)' % pre]
    for line in open(os.path.join(dst, pre + '_pmc_mfma.txt')):
        m = re.search(r'^(.*?)\s+n=\s*(\d+)\s+dur_ms\s+([\d.]+)\s+(.*)$', line)
        if not m:
            continue
        vals = dict((k, float(v)) for k, v in re.findall(r'(\w+)=([\d.e+]+)', m.group(4)))
        if vals.get('GRBM_GUI_ACTIVE', 0) > 0 and 'SQ_VALU_MFMA_BUSY_CYCLES' in vals:
            busy = vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * vals['GRBM_GUI_ACTIVE'] / 8)
            if busy > 0.02:
                lines.append('%-48s launches %4d  %7.3f ms  MFMA busy %5.1f %%' % (m.group(1)[:48], int(m.group(2)), float(m.group(3)), 100 * busy))
    open(os.path.join(dst, pre + '_mfma_busy_per_kernel.txt'), 'w').write('\n'.join(lines) + '\n')
    notes = subprocess.check_output([sys.executable, os.path.join(ROOT, 'tools', 'codeobj_notes.py')]).decode()
    open(os.path.join(dst, pre + '_codeobj_notes.txt'), 'w').write(notes)
    print('bench: %.2f evals/s, %.3f ms; LAUUM %.3f ms = %.1f %% of peak; traffic %.2f GB/launch (hash %s)' % (
        bench['value'], bench['ms_per_step'], bench['roofline']['launch_ms'], 100 * bench['roofline']['frac'],
        traffic['tile_gemm_lauum_bytes_per_launch'] / 1e9, lib_hash))


if __name__ == '__main__':
    main()
