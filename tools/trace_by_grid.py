"""Aggregate a rocprofv3 kernel trace (`*_kernel_trace.csv`) by (kernel, grid size, workgroup size): launches, average and
total duration per step - the per-shape view of the REPLAYED GRAPH (tools/shape_profile.py brackets an eager pass, where small
launches read long).  Also reports how much of the wall time two or more kernels were running at once (the branch streams).

    python tools/trace_by_grid.py /tmp/fsv_prof_raw/prof/p_kernel_trace.csv --steps 13 --out profiles/r02_trace_by_grid.jsonl
"""
import argparse
import collections
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=13)
    ap.add_argument('--out', default=None)
    ap.add_argument('--top', type=int, default=60)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    if not rows:
        raise SystemExit('empty trace')
    cols = rows[0].keys()
    pick = lambda *names: next(n for n in names if n in cols)
    kname = pick('Kernel_Name', 'kernel_name', 'Name')
    t0c, t1c = pick('Start_Timestamp', 'start_timestamp'), pick('End_Timestamp', 'end_timestamp')
    gx = pick('Grid_Size_X', 'grid_size_x', 'Grid_Size')
    wx = pick('Workgroup_Size_X', 'workgroup_size_x', 'Workgroup_Size')
    gy = 'Grid_Size_Y' if 'Grid_Size_Y' in cols else None
    gz = 'Grid_Size_Z' if 'Grid_Size_Z' in cols else None
    agg = collections.defaultdict(lambda: [0, 0.0])
    spans = []
    for r in rows:
        t0, t1 = int(r[t0c]), int(r[t1c])
        grid = (int(r[gx]), int(r[gy]) if gy else 1, int(r[gz]) if gz else 1)
        key = (r[kname].split('(')[0][:70], grid, int(r[wx]))
        agg[key][0] += 1
        agg[key][1] += (t1 - t0) * 1e-3
        spans.append((t0, t1))
    # overlap: sweep over start / end events
    ev = sorted([(t, 1) for t, _ in spans] + [(t, -1) for _, t in spans])
    busy1 = busy2 = 0
    depth, last = 0, ev[0][0]
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        depth += d
        last = t
    out = []
    for (name, grid, wg), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        wgs = (grid[0] // max(wg, 1)) * grid[1] * grid[2]
        out.append(dict(kernel=name, grid=list(grid), workgroup=wg, workgroups=wgs, launches_per_step=round(n / a.steps, 2),
                        avg_us=round(us / n, 2), ms_per_step=round(us / a.steps * 1e-3, 3)))
    print('kernel time %.2f ms/step, GPU busy %.2f ms/step, >= 2 kernels at once %.2f ms/step'
          % (sum(o['ms_per_step'] for o in out), busy1 * 1e-6 / a.steps, busy2 * 1e-6 / a.steps))
    for o in out[:a.top]:
        print('%7.3f ms  n=%6.1f  %8.1f us  wgs=%7d  %s' % (o['ms_per_step'], o['launches_per_step'], o['avg_us'], o['workgroups'],
                                                           o['kernel']))
    if a.out:
        with open(a.out, 'w') as f:
            f.write(json.dumps(dict(summary=True, gpu_busy_ms_per_step=round(busy1 * 1e-6 / a.steps, 3),
                                    overlapped_ms_per_step=round(busy2 * 1e-6 / a.steps, 3))) + '\n')
            for o in out:
                f.write(json.dumps(o) + '\n')


if __name__ == '__main__':
    main()
