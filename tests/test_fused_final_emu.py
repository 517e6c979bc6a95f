"""Opt-in fused second stage of the BatchNorm-shaped reductions (csrc/norm.hip fsv_red2f_kernel: the workgroup that finishes
last for a channel slab sums the slab's partials): bit-identical to the two-launch form at operator and training-step level, and
it removes the second-stage launches."""
import ctypes
import importlib

import torch

import model_checks as mc
import op_checks as oc

DEV = torch.device("cpu")


def _launches():
    lib = importlib.import_module('few-shot-vid2vid_amd.lib')
    fn = lib.get_lib().fsv_emu_launch_count
    fn.restype = ctypes.c_longlong
    return int(fn())


def test_operator_results_are_bit_identical(emu_lib):
    import fused_final_checks as fc
    fc.check_bit_identical(DEV, [(2, 12, 9, 7), (3, 130, 5, 6), (1, 7, 33, 31), (4, 64, 16, 16)], reps=2)


def test_operator_saves_one_launch_per_reduction(emu_lib):
    ops, conv = oc.pkg()
    x = torch.randn(2, 12, 9, 7)
    counts = []
    for fused in (False, True):
        prev = ops.set_fused_final(fused)
        try:
            xd = x.clone().requires_grad_(True)
            n0 = _launches()
            y = ops.norm_act(xd, None, None, torch.zeros(12), torch.ones(12), instance=False, training=True)
            y.backward(torch.ones_like(y))
            ops.colsum(conv.to_nhwc(x), 1, 2 * 9 * 7, 12)
            counts.append(_launches() - n0)
        finally:
            ops.set_fused_final(prev)
    assert counts[1] == counts[0] - 3, counts       # statistics, backward sums, column sums


def test_training_step_is_bit_identical_with_fewer_launches(emu_lib):
    ops, _ = oc.pkg()
    M = mc._model()
    out = []
    for fused in (False, True):
        prev = ops.set_fused_final(fused)
        try:
            opt = mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2)
            model = M.create_model(opt)
            mc.fill_state(model.netG); mc.fill_state(model.netD)
            model.train()
            opt_G, opt_D = model.build_optimizers()
            tl, ti, rl, ri = mc.synth_pose_inputs(1, 32, 32, 901, opt.input_nc)
            data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
            counts = []
            for it in range(2):
                n0 = _launches()
                d = M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
                g, _, _ = model(data, mode='generator')
                g = M.loss_backward(opt, g, opt_G, 0)
                counts.append(_launches() - n0)
            out.append((opt_G.flat_p.clone(), opt_D.flat_p.clone(), [float(x.detach()) for x in d],
                        [float(x.detach()) for x in g if not isinstance(x, int)], counts[1]))
        finally:
            ops.set_fused_final(prev)
    a, b = out
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    print('launches per iteration: two-stage %d, fused %d' % (a[4], b[4]))
    assert b[4] <= a[4] - 60, (a[4], b[4])
