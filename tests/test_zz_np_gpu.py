"""Hardware checks that run in their own process (a device fault there cannot take the test session down) and last in the
suite: the narrow-operand (`--amp`) kernels against their CPU definition, and GraphedIteration's hipGraph capture."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(script, marker, timeout):
    env = dict(os.environ)
    env.pop('FSV2V_EMU', None)
    r = subprocess.run([sys.executable, os.path.join(HERE, script)], cwd=HERE, env=env, capture_output=True, text=True,
                       timeout=timeout)
    sys.stdout.write(r.stdout[-6000:])
    sys.stderr.write(r.stderr[-6000:])
    assert r.returncode == 0 and marker in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_narrow_operand_kernels_on_hardware():
    _run('np_checks.py', 'NP_GPU_OK', 420)


@pytest.mark.gpu
def test_graphed_iteration_on_hardware():
    _run('graph_step_checks.py', 'GRAPH_STEP_GPU_OK', 300)
