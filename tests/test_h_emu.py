"""Half-precision convolution kernels (csrc/conv_h.hip) under the SIMT emulator: every tile / buffer count / split, data and
weight gradients, grouped launches, epilogue statistics, element conversion."""
import pytest
import torch

import h_checks as hc

DEV = torch.device("cpu")


def test_cast(emu_lib):
    hc.check_cast(DEV)


@pytest.mark.parametrize('gi', range(len(hc.GEOMS)))
@pytest.mark.parametrize('out_half', [True, False])
def test_forward_plan(emu_lib, gi, out_half):
    hc.check_forward(DEV, hc.GEOMS[gi], -1, 0, out_half)


@pytest.mark.parametrize('tile,split', hc.FWD_TILES)
def test_forward_tiles(emu_lib, tile, split):
    hc.check_forward(DEV, hc.GEOMS[1], tile, split, True)
    hc.check_forward(DEV, hc.GEOMS[4], tile, split, False, res_half=False)


def test_forward_half_residual_no_activation(emu_lib):
    hc.check_forward(DEV, hc.GEOMS[0], -1, 0, True, res_half=True, act=False)


@pytest.mark.parametrize('gi', range(len(hc.GEOMS)))
def test_dgrad(emu_lib, gi):
    hc.check_dgrad(DEV, hc.GEOMS[gi], True)
    hc.check_dgrad(DEV, hc.GEOMS[gi], False)


@pytest.mark.parametrize('tile,split', hc.WGRAD_TILES)
def test_wgrad_tiles(emu_lib, tile, split):
    assert hc.check_wgrad(DEV, hc.WG_GEOMS[0], tile, split)
    assert hc.check_wgrad(DEV, hc.WG_GEOMS[1], tile, split)


def test_wgrad_geometries(emu_lib):
    ran = [hc.check_wgrad(DEV, g, 0, 0) for g in hc.GEOMS + hc.WG_GEOMS]
    assert sum(ran) >= 5, ran


def test_group(emu_lib):
    hc.check_group(DEV)


def test_stats(emu_lib):
    hc.check_stats(DEV)


@pytest.mark.parametrize('nmaps,up', [(1, False), (3, True)])
def test_spade_half_output_and_half_gradient(emu_lib, nmaps, up):
    import op_checks as oc
    oc.check_spade(DEV, nmaps=nmaps, generated=True, c=32, ch=16, h=8, w=10, up=up, half_out=True)


@pytest.mark.parametrize('nmaps,up,c,ch,generated', [(1, False, 32, 16, True), (3, True, 64, 40, True), (2, False, 96, 72, False),
                                                     (1, True, 16, 8, True)])
def test_spade_f16_gemms(emu_lib, nmaps, up, c, ch, generated):
    """gamma / beta GEMMs of the fused modulation kernel and its backward twin on the f16 matrix instructions (csrc/spade.hip F16):
    both tile shapes (C <= 32: 128 x 32; else 64 x 64), K tails (Ch = 8, 40, 72: partial 32-wide chunks), partial channel tiles
    (C = 96 on 64-wide tiles), three maps, per-sample and shared weights"""
    import op_checks as oc
    oc.check_spade(DEV, nmaps=nmaps, generated=generated, c=c, ch=ch, h=8, w=10, up=up, half_out=True, f16=True)


@pytest.mark.parametrize('gx', [1, 2, 4])
@pytest.mark.parametrize('f16', [False, True])
def test_spade_workgroups_walk_several_pixel_tiles(emu_lib, monkeypatch, gx, f16):
    """a workgroup of the modulation kernels walks the pixel tiles blockIdx.x, + gridDim.x, ...; the chunk sequence (and the
    prefetch of x) runs across tile boundaries: 6 tiles of 64 pixels over 1 / 2 / 4 workgroups (even and uneven shares), three
    maps and one map (the two forms of the backward twin), with and without the folded up-sampling"""
    import op_checks as oc
    monkeypatch.setenv('FSV_SPADE_MAX_GX', str(gx))
    for nmaps, up, c in ((3, True, 64), (1, False, 64), (2, True, 32)):
        if f16:
            oc.check_spade(DEV, nmaps=nmaps, generated=True, c=c, ch=16, h=16, w=24, up=up, half_out=True, f16=True)
        else:
            oc.check_spade(DEV, nmaps=nmaps, generated=True, c=c, ch=16, h=16, w=24, up=up)


def test_spade_f16_bias_sums_over_copies(emu_lib, monkeypatch):
    """the bias sums of the backward twin spread over 4 copies of the buffer (what maps beyond 16 K pixels do)"""
    import op_checks as oc
    monkeypatch.setenv('FSV_SPADE_DB_SLOTS', '4')
    oc.check_spade(DEV, nmaps=2, generated=True, c=32, ch=16, h=16, w=24, up=False, half_out=True, f16=True)


def test_half_side_output_of_the_elementwise_producers(emu_lib):
    """under the half-precision kernels norm_act / its backward / act_backward also store their fp32 result as IEEE half
    (the explicit y_half / dx_half arguments of include/fsv2v.h): the copy equals the conversion pass it replaces and the consumer's to_half_nhwc takes
    it without a launch"""
    from importlib import import_module
    ops = import_module('few-shot-vid2vid_amd.ops')
    conv = import_module('few-shot-vid2vid_amd.conv')
    hconv = import_module('few-shot-vid2vid_amd.hconv')
    lib = emu_lib
    prev = conv.set_mfma_mode(1)
    try:
        assert conv.h_kernels()
        g = torch.Generator().manual_seed(3)
        x = conv.to_nhwc(torch.randn(2, 16, 5, 7, generator=g)).requires_grad_(True)
        w = torch.randn(16, generator=g).requires_grad_(True)
        b = torch.randn(16, generator=g).requires_grad_(True)
        y = ops.norm_act(x, w, b, None, None, instance=True, act=conv.ACT_LRELU)
        side = getattr(y, '_fsv_h16', None)
        assert side is not None and bool((side[1] == y.detach().to(torch.float16)).all())
        calls, real = [], lib.call

        def rec(name, *a):
            calls.append(name)
            return real(name, *a)
        lib.call = rec
        try:
            yh = hconv.to_half_nhwc(y)
        finally:
            lib.call = real
        assert yh is side[1] and 'fsv_cast_half' not in calls
        seen = {}
        y.register_hook(lambda gr: None)
        x.register_hook(lambda gr: seen.setdefault('dx', gr))
        dy = conv.to_nhwc(torch.randn(2, 16, 5, 7, generator=g))
        y.backward(dy)
        dx = seen['dx']
        side = getattr(dx, '_fsv_h16', None)
        assert side is not None and bool((side[1] == dx.to(torch.float16)).all())
        # odd channel counts / fp32 mode: no side output, the conversion pass runs as before
        x3 = conv.to_nhwc(torch.randn(1, 12, 4, 4, generator=g))
        assert getattr(ops.norm_act(x3, None, None, None, None, instance=True, act=conv.ACT_NONE), '_fsv_h16', None) is None
        d = ops.act_backward(dy, y.detach(), conv.ACT_LRELU)
        assert bool((d._fsv_h16[1] == d.to(torch.float16)).all())
    finally:
        conv.set_mfma_mode(prev)
    y2 = ops.norm_act(x.detach(), None, None, None, None, instance=True, act=conv.ACT_NONE)
    assert getattr(y2, '_fsv_h16', None) is None
