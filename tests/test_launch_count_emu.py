"""Library launches per training iteration (emulator counter): the eager step on the real stack is bound by the number of
launches, so the count of the steady-state iteration is pinned here (C3 flags, tiny widths - the count depends on the network's
structure, not on its size).  A change that adds launches has to raise the bound consciously."""
import ctypes
import importlib

import torch

import model_checks as mc

DEV = torch.device("cpu")


def _count():
    lib = importlib.import_module('few-shot-vid2vid_amd.lib')
    fn = lib.get_lib().fsv_emu_launch_count
    fn.restype = ctypes.c_longlong
    return int(fn())


def test_launches_per_iteration(emu_lib):
    M = mc._model()
    opt = mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True)
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    model.train()
    opt_G, opt_D = model.build_optimizers()
    tl, ti, rl, ri = mc.synth_pose_inputs(1, 64, 64, 900, opt.input_nc)
    data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
    counts = []
    for it in range(3):
        c0 = _count()
        M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
        g, _, _ = model(data, mode='generator')
        M.loss_backward(opt, g, opt_G, 0)
        counts.append(_count() - c0)
    print('library launches per iteration:', counts)
    # iteration 0 builds the layout tables and takes the un-grouped gradient paths; from iteration 1 on the count is steady
    assert counts[1] == counts[2], counts
    assert counts[2] <= counts[0], counts
    assert counts[2] <= 1850, counts          # 1805 when this bound was set (FSV_FUSED_FINAL=1: 1622)
