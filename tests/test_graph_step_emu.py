"""GraphedIteration's buffer plumbing (static inputs, signatures, outputs, device-side learning rate) on the emulator, where the
"graph" is an eager re-run on the static buffers: bit-identical to the plain loop of train.py:58-62."""
import torch

import graph_step_checks as gc

DEV = torch.device("cpu")


def test_graphed_iteration_matches_the_eager_loop(emu_lib):
    gc.check_graphed_iteration(DEV, iters=3)


def test_split_backward_without_exchange_keeps_every_gradient(emu_lib):
    gc.check_split_backward_single_rank(DEV, iters=2)


def test_capture_failure_falls_back_to_the_eager_step(emu_lib):
    gc.check_capture_failure_falls_back(DEV)


def test_discriminator_step_next_to_the_generator_pass_changes_no_result(emu_lib):
    gc.check_early_generator(DEV)


def test_decoder_stage_step_between_the_backward_pieces_changes_no_weight(emu_lib, monkeypatch):
    """FSV_EARLY_ADAM=1: Adam + layout refresh of flat_g[:split_at] on a side stream between the two pieces of a split backward
    (FlatAdam.step_stage2_early) - weights equal to the unsplit loop bit for bit, eager and graphed driver"""
    monkeypatch.setenv('FSV_EARLY_ADAM', '1')
    gc.check_split_backward_single_rank(DEV, iters=2)


def test_three_piece_backward_keeps_every_gradient(emu_lib):
    """build_optimizers(split_backward=3): a second stage boundary behind the reference encoders - weights equal to the one-piece
    loop bit for bit, eager and graphed driver"""
    gc.check_split_backward_single_rank(DEV, iters=2, pieces=3)


def test_twin_generator_passes_change_no_result_and_no_buffer(emu_lib):
    """round 6: the no-grad and the generator-mode pass of an iteration issued next to each other - plain loop and graphed driver"""
    gc.check_twin_generator_passes(DEV, iters=2)
    gc.check_twin_generator_passes(DEV, graphed=True)


def test_serial_point_opt_ins_change_no_bit(emu_lib):
    """FSV_LOSS_TICKET=1 on the emulated kernels (FSV_ZERO_EARLY=1 needs a device: the hardware run of graph_step_checks.py)"""
    gc.check_serial_point_opt_ins(DEV, iters=3)
