"""First hardware run of the code written after the round-1 GPU budget was spent: the narrow-operand (`--amp`) kernels and the
experimental fp32 tile variants (few-wave workgroups, double-buffered LDS; force_tile only, never picked by the launch plan).
Both cross-compile for gfx950 and pass the emulator suite (test_np_emu.py, test_amp_emu.py, test_tiles_emu.py) but have not
executed on an MI355X yet.  The checks therefore run in subprocesses (a fault there cannot take the test session down), last in
the suite, and the tests are non-strict xfails until a GPU run has confirmed them: XPASS in the summary = validated on hardware."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.xfail(strict=False, reason="narrow-operand kernels not yet validated on MI355X (emulator-verified only)")
def test_narrow_operand_kernels_on_hardware():
    env = dict(os.environ)
    env.pop('FSV2V_EMU', None)
    r = subprocess.run([sys.executable, os.path.join(HERE, 'np_checks.py')], cwd=HERE, env=env, capture_output=True, text=True,
                       timeout=420)
    sys.stdout.write(r.stdout[-4000:])
    sys.stderr.write(r.stderr[-4000:])
    assert r.returncode == 0 and 'NP_GPU_OK' in r.stdout


@pytest.mark.gpu
@pytest.mark.xfail(strict=False, reason="experimental tile variants not yet validated on MI355X (emulator-verified only)")
def test_experimental_tiles_on_hardware():
    env = dict(os.environ)
    env.pop('FSV2V_EMU', None)
    r = subprocess.run([sys.executable, os.path.join(HERE, 'tile_checks.py')], cwd=HERE, env=env, capture_output=True, text=True,
                       timeout=240)
    sys.stdout.write(r.stdout[-4000:])
    sys.stderr.write(r.stderr[-4000:])
    assert r.returncode == 0 and 'TILES_GPU_OK' in r.stdout


@pytest.mark.gpu
@pytest.mark.xfail(strict=False, reason="GraphedIteration's hipGraph capture not yet validated on MI355X (plumbing emulator-verified)")
def test_graphed_iteration_on_hardware():
    env = dict(os.environ)
    env.pop('FSV2V_EMU', None)
    r = subprocess.run([sys.executable, os.path.join(HERE, 'graph_step_checks.py')], cwd=HERE, env=env, capture_output=True,
                       text=True, timeout=300)
    sys.stdout.write(r.stdout[-4000:])
    sys.stderr.write(r.stderr[-4000:])
    assert r.returncode == 0 and 'GRAPH_STEP_GPU_OK' in r.stdout


@pytest.mark.gpu
@pytest.mark.xfail(strict=False, reason="fused reduction second stage not yet validated on MI355X (emulator-verified only)")
def test_fused_final_on_hardware():
    env = dict(os.environ)
    env.pop('FSV2V_EMU', None)
    r = subprocess.run([sys.executable, os.path.join(HERE, 'fused_final_checks.py')], cwd=HERE, env=env, capture_output=True,
                       text=True, timeout=180)
    sys.stdout.write(r.stdout[-4000:])
    sys.stderr.write(r.stderr[-4000:])
    assert r.returncode == 0 and 'FUSED_FINAL_GPU_OK' in r.stdout
