"""The half-precision path (`--amp O1`: csrc/conv_h.hip, the F16 forms of csrc/spade.hip) on a real MI355X, through the C ABI:
every tile / buffer count / K split of the gather-GEMM, data and weight gradients, grouped launches, epilogue statistics, the
SPADE modulation kernels with f16 GEMMs, and the whole `--amp` iteration of a small network against the whole-iteration oracle
in the same arithmetic.  The full-size configuration (BASELINE.json configs[4]) is tests/test_fullsize_gpu.py."""
import pytest
import torch

import h_checks as hc
import model_checks as mc
import op_checks as oc

pytestmark = pytest.mark.gpu

DEV = torch.device('cuda:0')


def test_cast(hip_lib):
    hc.check_cast(DEV)


@pytest.mark.parametrize('gi', range(len(hc.GEOMS + hc.BIG_GEOMS)))
def test_forward_dgrad_wgrad_planned(hip_lib, gi):
    geom = (hc.GEOMS + hc.BIG_GEOMS)[gi]
    for half in (True, False):
        hc.check_forward(DEV, geom, -1, 0, half)
        hc.check_dgrad(DEV, geom, half)
    hc.check_wgrad(DEV, geom, 0, 0)


@pytest.mark.parametrize('tile,split', hc.FWD_TILES)
def test_forward_tiles(hip_lib, tile, split):
    hc.check_forward(DEV, hc.BIG_GEOMS[0], tile, split, True)
    hc.check_forward(DEV, hc.BIG_GEOMS[2], tile, split, False, res_half=True)


@pytest.mark.parametrize('tile,split', hc.WGRAD_TILES)
def test_wgrad_tiles(hip_lib, tile, split):
    assert hc.check_wgrad(DEV, hc.BIG_GEOMS[1], tile, split)


def test_group_and_stats(hip_lib):
    hc.check_group(DEV)
    hc.check_stats(DEV)


@pytest.mark.parametrize('nmaps,up,c,ch,generated,hw', [(1, False, 32, 16, True, (8, 10)), (3, True, 64, 40, True, (8, 10)),
                                                        (2, False, 96, 72, False, (8, 10)), (1, True, 16, 8, True, (8, 10)),
                                                        (3, True, 128, 128, True, (64, 96)), (2, False, 32, 32, True, (96, 128))])
def test_spade_f16_gemms(hip_lib, nmaps, up, c, ch, generated, hw):
    oc.check_spade(DEV, nmaps=nmaps, generated=generated, c=c, ch=ch, h=hw[0], w=hw[1], up=up, half_out=True, f16=True)


@pytest.mark.parametrize('gx', [2, 4])
@pytest.mark.parametrize('f16', [False, True])
def test_spade_workgroups_walk_several_pixel_tiles(hip_lib, monkeypatch, gx, f16):
    monkeypatch.setenv('FSV_SPADE_MAX_GX', str(gx))
    for nmaps, up, c in ((3, True, 64), (1, False, 64), (2, True, 32)):
        if f16:
            oc.check_spade(DEV, nmaps=nmaps, generated=True, c=c, ch=16, h=16, w=24, up=up, half_out=True, f16=True)
        else:
            oc.check_spade(DEV, nmaps=nmaps, generated=True, c=c, ch=16, h=16, w=24, up=up)


def test_spade_large_map_persistent_grid(hip_lib):
    """more pixel tiles than resident workgroups (the production regime at 512 K pixels): 131072 pixels, C = 64"""
    oc.check_spade(DEV, nmaps=1, generated=True, n=1, c=64, ch=32, h=256, w=512, up=True, half_out=True, f16=True)
    oc.check_spade(DEV, nmaps=3, generated=True, n=1, c=64, ch=32, h=256, w=512, up=True)


def test_spade_f16_bias_sums_over_copies(hip_lib, monkeypatch):
    monkeypatch.setenv('FSV_SPADE_DB_SLOTS', '4')
    oc.check_spade(DEV, nmaps=2, generated=True, c=32, ch=16, h=16, w=24, up=False, half_out=True, f16=True)
    monkeypatch.delenv('FSV_SPADE_DB_SLOTS')
    oc.check_spade(DEV, nmaps=2, generated=True, c=32, ch=32, h=160, w=192, up=True, half_out=True, f16=True)      # 30720 pixels: 2 copies


@pytest.mark.parametrize('nmaps,up', [(1, False), (3, True)])
def test_spade_half_output_fp32_gemms(hip_lib, nmaps, up):
    oc.check_spade(DEV, nmaps=nmaps, generated=True, c=32, ch=16, h=8, w=10, up=up, half_out=True)


def test_amp_step_is_the_stated_definition_small(hip_lib):
    """as tests/test_amp_emu.py, on hardware: whole --amp O1 iteration against the whole-iteration oracle in the same arithmetic"""
    opt = mc.tiny_opt(dataset_mode='fewshot_street', label_nc=35, input_nc=3, aspect_ratio=2.0, fineSize=64, loadSize=64, batchSize=1,
                      amp='O1', ngf=16, ndf=16, nff=16, n_downsample_G=3, n_adaptive_layers=2)
    mc.check_amp_train_step(DEV, opt, b=1, tol=1e-3, grad_tol=2e-2)
