"""Every opt-in switch at once (experimental tiles swapped into the plan, double-buffered weight gradients, fused reduction
finals, merged stride-2 data gradients, workspace split-K): a training iteration still gives the default path's losses and
images.  Runs in subprocesses because the tile re-map is read from the environment once per process."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SWITCHES = dict(FSV_TILE_REMAP='4:16,9:17,1:18', FSV_WGRAD_VARIANT='db', FSV_FUSED_FINAL='1', FSV_DGRAD_MERGE='2', FSV_SPLITK_WS='2')


def _probe(extra):
    env = dict(os.environ, FSV2V_EMU='1', **extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, 'all_switches_probe.py')], cwd=HERE, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_all_switches_together(emu_lib):
    base, allon = _probe({}), _probe(SWITCHES)
    for k in ('d', 'g'):
        for a, b in zip(base[k], allon[k]):
            assert abs(a - b) <= 2e-4 * max(abs(a), 1.0), (k, base[k], allon[k])
    worst = max(abs(a - b) for a, b in zip(base['img'], allon['img']))
    assert worst <= 2e-4, worst
