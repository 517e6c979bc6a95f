"""Digest of gpurun_out/first_hw/ (written by tools/first_hw_pass.sh): test verdicts, whole-step numbers per variant, best tile per
layer shape, operand-mode speed-ups.  python tools/summarize_first_hw.py [dir]"""
import glob
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'first_hw')


def lines(name):
    p = os.path.join(d, name + '.log')
    return open(p).read().splitlines() if os.path.exists(p) else []


def jsonl(name):
    out = []
    for l in lines(name):
        l = l.strip()
        if l.startswith('{'):
            try:
                out.append(json.loads(l))
            except ValueError:
                pass
    return out


for name in ('pytest_new', 'pytest_all'):
    ls = [l for l in lines(name) if 'passed' in l or 'failed' in l or 'XPASS' in l or 'XFAIL' in l or 'error' in l.lower()]
    print('%-12s %s' % (name, ' | '.join(ls[-4:]) if ls else '(no output)'))
print()
base = None
for p in sorted(glob.glob(os.path.join(d, 'bench_*.log'))):
    name = os.path.basename(p)[:-4]
    r = jsonl(name)
    if not r:
        print('%-14s (no JSON line) %s' % (name, (lines(name) or [''])[-1][:120]))
        continue
    r = r[-1]
    if name == 'bench_f32':
        base = r['ms_per_step']
    rl = r.get('roofline') or {}
    print('%-14s %7.2f ms/step  %6.2f frames/s  %s  dominant %s %.1f TF/s (frac %.3f)' % (
        name, r['ms_per_step'], r['value'], ('x%.2f vs f32' % (base / r['ms_per_step'])) if base else '', rl.get('kernel', '-'),
        rl.get('achieved', 0.0), rl.get('frac', 0.0)))
print()
for name in ('tile_ab', 'wgrad_ab'):
    for row in jsonl(name):
        case = row.pop('case')
        auto = row.get('auto')
        best = max(row.items(), key=lambda kv: kv[1])
        exp = {k: v for k, v in row.items() if any(t in k for t in ('t10', 't11', 't12', 't13', 't14', 't15', 't16', 't17', 't18', 't19', 't20', 't21', 'fw', 'db'))}
        bexp = max(exp.items(), key=lambda kv: kv[1]) if exp else ('-', 0.0)
        print('%-9s %-24s auto %6.1f  best %-12s %6.1f  best experimental %-12s %6.1f' % (name, case, auto or 0.0, best[0], best[1], bexp[0], bexp[1]))
    print()
for row in jsonl('np_ab'):
    print('np_ab     %-20s fwd f32 %6.1f  f16 %7.1f (x%.1f, err %.1e)  bf16x3 %7.1f (x%.1f, err %.1e) | wgrad f32 %6.1f  f16 %7.1f  bf16x3 %7.1f' % (
        row['case'], row['fwd_f32'], row['fwd_f16'], row['fwd_f16'] / row['fwd_f32'], row.get('fwd_f16_err', 0), row['fwd_bf16x3'],
        row['fwd_bf16x3'] / row['fwd_f32'], row.get('fwd_bf16x3_err', 0), row['wgrad_f32'], row['wgrad_f16'], row['wgrad_bf16x3']))
