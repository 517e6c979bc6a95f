"""GraphedIteration's buffer plumbing (static inputs, signatures, outputs, device-side learning rate) on the emulator, where the
"graph" is an eager re-run on the static buffers: bit-identical to the plain loop of train.py:58-62."""
import torch

import graph_step_checks as gc

DEV = torch.device("cpu")


def test_graphed_iteration_matches_the_eager_loop(emu_lib):
    gc.check_graphed_iteration(DEV, iters=3)
