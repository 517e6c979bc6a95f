"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.

usage: python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [workload [amp]]
Both counters are reported in KiB per dispatch.  On gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
read (MI355X_MICROARCH.md, HBM section): the read figure is doubled ("read_MB_corrected"); WRITE_SIZE is taken as is."""
import csv, glob, json, os, sys


def load(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r['Counter_Name'] != counter:
                    continue
                name = r['Kernel_Name'].split('(')[0].strip()
                a = out.setdefault(name, [0, 0.0])
                a[0] += 1
                a[1] += float(r['Counter_Value'])
    return out


def main():
    fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    res = {}
    for name, (n, tot) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
        w = write.get(name, [1, 0.0])
        res[name] = dict(launches=n, fetch_kib_raw_per_launch=round(tot / n, 1),
                         read_MB_corrected=round(2 * tot / n * 1024 / 1e6, 2),
                         write_MB=round(w[1] / max(w[0], 1) * 1024 / 1e6, 2))
    # which build these counters describe: the digest of the kernel sources (few-shot-vid2vid_amd/build.py source_digest) and,
    # where a git checkout is at hand, the commit - bench.py only quotes `roofline.traffic` from a file whose digest matches
    # the library it is running
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import fsv2v_amd  # noqa: F401
    from importlib import import_module
    meta = dict(source_digest=import_module('few-shot-vid2vid_amd.build').source_digest())
    try:
        import subprocess
        meta['commit'] = subprocess.run(['git', 'rev-parse', 'HEAD'], capture_output=True, text=True, timeout=5,
                                        cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
    except Exception:
        meta['commit'] = None
    if not meta['commit']:
        # the GPU box's snapshot has no .git: the commit the snapshot was cut from travels in .build_commit (written in the build
        # container right before the gpurun call, `git rev-parse HEAD`, suffixed "+dirty" when the tree had uncommitted changes)
        try:
            with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '.build_commit')) as f:
                meta['commit'] = f.read().strip() or None
        except OSError:
            pass
    meta['workload'] = sys.argv[4] if len(sys.argv) > 4 else 'pose'
    meta['amp'] = sys.argv[5] if len(sys.argv) > 5 else 'O0'
    with open(sys.argv[3], 'w') as f:
        json.dump(dict(_build=meta, **res), f, indent=1)
    for k in list(res)[:12]:
        print(k[:70], res[k])


if __name__ == '__main__':
    main()
