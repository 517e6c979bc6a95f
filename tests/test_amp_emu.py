"""`--amp` end to end under the SIMT emulator: operand-mode selection from the option string, the loss scale through
loss_backward and the fused step, overflow handling against apex's rule (oracle/np_oracle.LossScaler)."""
import pytest
import torch

import model_checks as mc

DEV = torch.device("cpu")


def test_amp_level_parsing(emu_lib):
    M = mc._model()
    assert M.amp_mode(mc.tiny_opt()) == 0                   # the reference's default 'O0'
    assert M.amp_mode(mc.tiny_opt(amp='O1')) == 1
    assert M.amp_mode(mc.tiny_opt(amp='O2')) == 1
    assert M.amp_mode(mc.tiny_opt(amp='bf16x3')) == 2
    with pytest.raises(ValueError):
        M.amp_mode(mc.tiny_opt(amp='O9'))


def test_overflow_skips_step_and_adapts_scale(emu_lib):
    mc.check_amp_overflow_skip(DEV)


def test_train_step_f16_operands_tiny(emu_lib):
    mc.check_amp_step(DEV, dict(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32,
                                n_downsample_G=3, n_adaptive_layers=2), 'O1')


def test_train_step_bf16x3_operands_tiny(emu_lib):
    mc.check_amp_step(DEV, dict(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32,
                                n_downsample_G=3, n_adaptive_layers=2), 'bf16x3', loss_tol=1e-3, image_tol=1e-3, grad_l2_tol=2e-2)


def test_amp_step_is_the_stated_definition_small(emu_lib):
    """the whole --amp O1 iteration of the product against the whole-iteration oracle in the same arithmetic (model_checks.
    check_amp_train_step) at a width where every layer class takes the half-precision kernels (channels multiples of 8)"""
    opt = mc.tiny_opt(dataset_mode='fewshot_street', label_nc=35, input_nc=3, aspect_ratio=2.0, fineSize=64, loadSize=64, batchSize=1,
                      amp='O1', ngf=16, ndf=16, nff=16, n_downsample_G=3, n_adaptive_layers=2)
    mc.check_amp_train_step(DEV, opt, b=1, tol=1e-3, grad_tol=2e-2)


def test_amp_step_is_the_stated_definition_small_in_the_schedule_bench_py_runs(emu_lib):
    """the whole --amp O1 iteration of the product against the whole-iteration oracle in the same arithmetic (model_checks.
    check_amp_train_step) at a width where every layer class takes the half-precision kernels (channels multiples of 8)"""
    opt = mc.tiny_opt(dataset_mode='fewshot_street', label_nc=35, input_nc=3, aspect_ratio=2.0, fineSize=64, loadSize=64, batchSize=1,
                      amp='O1', ngf=16, ndf=16, nff=16, n_downsample_G=3, n_adaptive_layers=2)
    mc.check_amp_train_step(DEV, opt, b=1, tol=1e-3, grad_tol=2e-2, bench_schedule=True)
