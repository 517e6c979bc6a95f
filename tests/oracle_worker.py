"""Computes whole-iteration oracle pairs AHEAD of the full-size hardware tests that need them (TEST INFRASTRUCTURE ONLY).

    python tests/oracle_worker.py <cache dir> <threads> <spec file (JSON list)> [comma-separated cores to pin to]

Started by tests/conftest.py next to a `-m gpu` session that contains tests/test_fullsize_gpu.py, when FSV_ORACLE_WORKER=1 (opt-in: two
OpenMP processes on the same cores spin against each other - conftest.py pins this one to cores of its own; the measurements are in
the comment there): the (fp32, fp64) oracle runs of
the benchmarked configurations are minutes of host time each, and nothing about them needs the GPU - while it runs the rest of the
suite, this process works through the specs in the order the tests will ask for them and leaves each pair in the cache directory
(model_checks.oracle_pair picks it up; a pair that is not ready yet is waited for, a failed one is computed inline by the test, so
the worker can never cost a test its result).  The oracle itself is untouched: same functions, same inputs, same process-external
state as the inline call."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    cdir, threads, spec_file = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    if len(sys.argv) > 4 and sys.argv[4]:
        os.sched_setaffinity(0, {int(c) for c in sys.argv[4].split(',')})          # before the OpenMP runtime starts its threads
    os.environ.pop('FSV2V_EMU', None)
    os.environ['CUDA_VISIBLE_DEVICES'] = ''          # the oracle is host code; never touch the device the tests are using
    os.environ['HIP_VISIBLE_DEVICES'] = ''
    import torch
    torch.set_num_threads(max(threads, 1))
    import model_checks as mc
    specs = json.load(open(spec_file))
    with open(os.path.join(cdir, 'worker.pid'), 'w') as f:
        f.write(str(os.getpid()))
    keys = [mc.oracle_key(s) for s in specs]
    for k in keys:
        open(os.path.join(cdir, k + '.queued'), 'w').close()
    for spec, k in zip(specs, keys):
        try:
            pair = mc.compute_oracle_pair(spec)
            tmp = os.path.join(cdir, k + '.tmp')
            torch.save(pair, tmp)
            os.replace(tmp, os.path.join(cdir, k + '.pt'))
            del pair
        except Exception as e:                       # noqa: BLE001 - the test computes inline
            print('oracle worker: %s failed: %s' % (k, str(e).split('\n')[0]), file=sys.stderr, flush=True)
        finally:
            try:
                os.remove(os.path.join(cdir, k + '.queued'))
            except OSError:
                pass


if __name__ == '__main__':
    main()
