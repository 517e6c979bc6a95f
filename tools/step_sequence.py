"""One replayed step of a rocprofv3 kernel trace as an ordered launch list: start offset, duration, queue, short kernel name, grid.

The aggregate views (kernel stats, trace_by_grid) say how much time a kernel takes; this one says WHERE the small launches sit -
which torch-native copies / fills / adds surround which fsv kernel - and how the queues (streams) interleave.

    python tools/step_sequence.py /tmp/fsv_prof_raw/prof/p_kernel_trace.csv --steps 13 --which 10 --out profiles/r06_step_sequence.txt
"""
import argparse
import csv
import re


def short(name):
    name = name.split('(')[0]
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'at::native::(\(anonymous namespace\)::)?', 'at::', name)
    return name[:96]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=13, help='replays + eager passes in the trace (the last `steps` are equal-length groups)')
    ap.add_argument('--which', type=int, default=-2, help='which of the equal groups to print (default: the one before the last)')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    cols = rows[0].keys()
    pick = lambda *names: next(n for n in names if n in cols)
    kname, t0c, t1c = pick('Kernel_Name', 'Name'), pick('Start_Timestamp'), pick('End_Timestamp')
    qc = pick('Queue_Id', 'Stream_Id')
    rows.sort(key=lambda r: int(r[t0c]))
    # the timed replays are the tail of the trace: find the period by the Adam kernel of the generator (the last big fsv_adam launch of a step)
    marks = [i for i, r in enumerate(rows) if r[kname].startswith('fsv_adam_kernel')]
    per = 2                                   # Adam(D), Adam(G) per step
    ends = marks[per - 1::per]
    if len(ends) < 3:
        raise SystemExit('no step structure found')
    w = a.which if a.which >= 0 else len(ends) + a.which
    lo, hi = ends[w - 1] + 1, ends[w] + 1
    step = rows[lo:hi]
    # everything up to the end of the layout refresh behind the last Adam belongs to the step: extend to the next D-forward start
    base = int(step[0][t0c])
    lines = []
    tiny = 0
    for r in step:
        t0, t1 = int(r[t0c]), int(r[t1c])
        d = (t1 - t0) * 1e-3
        tiny += d < 8.0
        gx = r.get('Grid_Size_X', r.get('Grid_Size', '0'))
        wx = r.get('Workgroup_Size_X', r.get('Workgroup_Size', '1'))
        wgs = int(gx) // max(int(wx), 1) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
        lines.append('%9.1f us  %7.1f us  q%-3s wgs=%-6d %s' % ((t0 - base) * 1e-3, d, r[qc], wgs, short(r[kname])))
    span = (int(step[-1][t1c]) - base) * 1e-6
    head = '# %d launches, %.3f ms from first start to last end, %d launches shorter than 8 us' % (len(step), span, tiny)
    text = head + '\n' + '\n'.join(lines) + '\n'
    if a.out:
        open(a.out, 'w').write(text)
    print(head)


if __name__ == '__main__':
    main()
