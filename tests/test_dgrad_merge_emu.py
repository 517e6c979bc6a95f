"""Opt-in merged stride-2 data gradient (csrc/conv_igemm_db.hip fsv_conv_igemm_db4_kernel: the four output-parity classes of a
layer in one launch): bit-identical to the per-class launches, including odd sizes (classes of different extent), 3x3 and 4x4
kernels, with and without the persistent layouts; it declines (and the loop takes over) where its tile has no variant."""
import ctypes
import importlib

import pytest
import torch

import op_checks as oc

DEV = torch.device("cpu")


def _launches():
    lib = importlib.import_module('few-shot-vid2vid_amd.lib')
    fn = lib.get_lib().fsv_emu_launch_count
    fn.restype = ctypes.c_longlong
    return int(fn())


GEOMS = [  # n, cin, h, w, cout, k, pad
    (2, 12, 11, 13, 132, 3, 1), (1, 20, 10, 9, 40, 4, 2), (2, 64, 16, 16, 64, 3, 1), (1, 16, 7, 8, 24, 4, 1), (3, 8, 9, 9, 72, 3, 1),
]


def _run(geom, merge):
    """True when the merged launch ran (False: the plan wanted split-K, or the entry point declined its tile)"""
    ops, conv = oc.pkg()
    n, cin, h, w, cout, k, p = geom
    g = torch.Generator().manual_seed(hash(geom) % 1000)
    geo = conv.Geom(k, k, 2, p)
    oh, ow = geo.out_hw(h, w)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.2
    dout = conv.to_nhwc(torch.randn(n, cout, oh, ow, generator=g))
    res, counts = [], []
    for m in (0, merge):
        prev = conv.set_dgrad_merge(m)
        try:
            n0 = _launches()
            res.append(conv.conv_dgrad(dout, wt, geo, (h, w)))
            counts.append(_launches() - n0)
        finally:
            conv.set_dgrad_merge(prev)
    assert torch.equal(res[0], res[1])
    # 4 weight re-arrangements + 4 GEMMs -> 4 + 1 when the merged launch ran; unchanged when the per-class plan wants split-K (the
    # merged path is not even tried); + 4 unused re-arrangements when the entry point declined its tile
    assert counts[1] in (counts[0] - 3, counts[0], counts[0] + 4), counts
    return counts[1] == counts[0] - 3


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("merge", [1, 2])
def test_merged_dgrad_is_bit_identical(emu_lib, geom, merge):
    _run(geom, merge)


def test_merged_path_is_actually_taken(emu_lib):
    taken = [_run(g, 1) for g in GEOMS + [(2, 64, 32, 32, 96, 3, 1), (2, 128, 24, 24, 64, 4, 1)]]
    assert sum(taken) >= 2, taken


def test_step_is_bit_identical_with_fewer_launches(emu_lib):
    import model_checks as mc
    ops, conv = oc.pkg()
    M = mc._model()
    out = []
    for m in (0, 1):
        prev = conv.set_dgrad_merge(m)
        try:
            opt = mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, ngf=16, nff=16, fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2)
            model = M.create_model(opt)
            mc.fill_state(model.netG); mc.fill_state(model.netD)
            model.train()
            opt_G, opt_D = model.build_optimizers()
            tl, ti, rl, ri = mc.synth_pose_inputs(1, 32, 32, 902, opt.input_nc)
            data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
            for it in range(2):
                n0 = _launches()
                d = M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
                g, _, _ = model(data, mode='generator')
                g = M.loss_backward(opt, g, opt_G, 0)
                cnt = _launches() - n0
            out.append((opt_G.flat_p.clone(), opt_D.flat_p.clone(), cnt))
        finally:
            conv.set_dgrad_merge(prev)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    print('launches per iteration: per-class %d, merged %d' % (out[0][2], out[1][2]))
    assert out[1][2] <= out[0][2]
