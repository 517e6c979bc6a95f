"""Opt-in split-K through a workspace (csrc/conv_igemm_db.hip, WSK): one launch instead of zero fill + atomic GEMM + finishing
pass, fixed summation order.  Checked against torch and against the shipped split-K path (same values up to the fp32 summation
order), at operator and training-step level, with the launch counter."""
import ctypes
import importlib

import pytest
import torch
import torch.nn.functional as F

import op_checks as oc

DEV = torch.device("cpu")


def _launches():
    lib = importlib.import_module('few-shot-vid2vid_amd.lib')
    fn = lib.get_lib().fsv_emu_launch_count
    fn.restype = ctypes.c_longlong
    return int(fn())


GEOMS = [  # n, cin, h, w, cout, k, stride, pad   (small pixel counts, long K: the plan splits)
    (2, 256, 8, 8, 256, 3, 1, 1), (1, 512, 4, 4, 192, 3, 1, 1), (2, 128, 9, 7, 72, 3, 1, 1), (2, 1024, 4, 4, 64, 1, 1, 0),
    (1, 256, 6, 6, 130, 4, 2, 1),
]


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("mode", [1, 2])
def test_workspace_split_matches_torch_and_saves_launches(emu_lib, geom, mode):
    ops, conv = oc.pkg()
    n, cin, h, w, cout, k, s, p = geom
    g = torch.Generator().manual_seed(abs(hash(geom)) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, *conv.Geom(k, k, s, p).out_hw(h, w), generator=g)
    ref = F.leaky_relu(F.conv2d(x, wt, b, stride=s, padding=p), 0.2) + res
    geo = conv.Geom(k, k, s, p)
    wf, _, ldw = conv.prep_weight(wt, 0, geo)
    outs, counts = [], []
    for m in (0, mode):
        prev = conv.set_splitk_ws(m)
        try:
            n0 = _launches()
            outs.append(conv.conv_forward(conv.to_nhwc(x), wf, ldw, cout, geo, bias=b, res=res, act=conv.ACT_LRELU))
            counts.append(_launches() - n0)
        finally:
            conv.set_splitk_ws(prev)
    oc.assert_close('split-K (atomics)', outs[0], ref, 1e-4)
    oc.assert_close('split-K (workspace)', outs[1], ref, 1e-4)
    oc.assert_close('workspace vs atomics', outs[1], outs[0], 1e-5)
    assert counts == [2, 1], counts          # GEMM + finishing pass (the zero fill is a memset, not a kernel of the library) -> one


def test_step_matches_and_saves_launches(emu_lib):
    import model_checks as mc
    ops, conv = oc.pkg()
    M = mc._model()
    out = []
    for m in (0, 1):
        prev = conv.set_splitk_ws(m)
        try:
            opt = mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, ngf=16, nff=16, ndf=8, fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2)
            model = M.create_model(opt)
            mc.fill_state(model.netG); mc.fill_state(model.netD)
            model.train()
            opt_G, opt_D = model.build_optimizers()
            opt_G.set_lr(0.0); opt_D.set_lr(0.0)
            tl, ti, rl, ri = mc.synth_pose_inputs(1, 32, 32, 903, opt.input_nc)
            data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
            for it in range(2):
                n0 = _launches()
                d = M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
                g, gen, _ = model(data, save_images=True, mode='generator')
                g = M.loss_backward(opt, g, opt_G, 0)
                cnt = _launches() - n0
            out.append(([float(x.detach()) for x in d] + [float(x.detach()) for x in g if not isinstance(x, int)], gen[0].detach().clone(), cnt))
        finally:
            conv.set_splitk_ws(prev)
    for a, b in zip(out[0][0], out[1][0]):
        assert abs(a - b) <= 1e-4 * max(abs(a), 1.0), (out[0][0], out[1][0])
    assert float((out[0][1] - out[1][1]).abs().max()) <= 1e-4
    print('launches per iteration: atomics %d, workspace %d' % (out[0][2], out[1][2]))
    assert out[1][2] < out[0][2]
