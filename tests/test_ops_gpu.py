"""Operator parity on a real MI355X: HIP kernels (through the C ABI) vs the CPU oracle on identical inputs."""
import pytest
import torch

import op_checks as oc

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("cfg", [
    (1, 8, 8, 8, 8, 3, 1, 1, 'lrelu', True),
    (2, 4, 9, 7, 5, 3, 2, 1, 'none', True),
    (1, 6, 10, 10, 12, 3, 1, 1, 'tanh', True),
    (1, 20, 9, 9, 32, 4, 2, 2, 'lrelu', True),
    (1, 8, 7, 7, 16, 4, 1, 2, 'sigmoid', False),
    (1, 64, 6, 6, 130, 1, 1, 0, 'lrelu', True),
    (1, 32, 9, 9, 48, 1, 1, 0, 'none', True),        # small-K weight-gradient tiles (32 / 64 rows)
    (1, 64, 9, 9, 64, 1, 1, 0, 'none', True),
    (2, 32, 5, 7, 130, 1, 1, 0, 'none', False),
    (2, 64, 64, 64, 32, 3, 1, 1, 'lrelu', True),      # up_0 conv_0 shape at reduced resolution
    (2, 128, 32, 32, 256, 3, 2, 1, 'none', True),     # stride-2 encoder conv
    (2, 512, 8, 8, 512, 3, 1, 1, 'none', True),       # split-K regime (few tiles, long K)
    (2, 20, 65, 65, 32, 4, 2, 2, 'lrelu', True),      # PatchGAN odd geometry
    (1, 15, 40, 24, 32, 3, 1, 1, 'lrelu', True),      # FlowGenerator first conv (15 input channels)
])
def test_conv(hip_lib, cfg):
    n, cin, h, w, cout, k, s, p, act, bias = cfg
    oc.check_conv(dev(), n, cin, h, w, cout, k, s, p, act=act, bias=bias)


def test_conv_spectral_residual(hip_lib):
    oc.check_conv_sn_res(dev())


def test_layout_cache(hip_lib):
    oc.check_layout_cache(dev())


def test_deferred_wgrad_finalize(hip_lib):
    oc.check_deferred_wgrad(dev())


def test_linear(hip_lib):
    oc.check_linear(dev())
    oc.check_linear(dev(), r=2048, cin=256, cout=514)


def test_batch_conv(hip_lib):
    oc.check_batch_conv(dev())
    oc.check_batch_conv(dev(), b=2, cin=64, cout=32, h=32, w=32)


@pytest.mark.parametrize("instance,affine,act", [(False, True, 'lrelu'), (True, True, 'lrelu'), (False, False, 'none')])
def test_norm(hip_lib, instance, affine, act):
    oc.check_norm(dev(), instance=instance, affine=affine, act=act)
    oc.check_norm(dev(), instance=instance, affine=affine, act=act, n=2, c=64, h=65, w=33)


def test_spectral_power_iteration_is_reproducible(hip_lib):
    """same bits on every run (25 runs of three layers up to 512 x 4608, single-layer and grouped entry)"""
    oc.check_spectral_power_iteration(dev(), repeats=25)


def test_cat_and_pad_forms(hip_lib):
    oc.check_cat_and_pad(dev())


def test_fused_reductions(hip_lib):
    """last-workgroup second stage across the 8 XCDs: many repeats on the same buffers, sizes on both sides of the threshold"""
    oc.check_fused_reductions(dev(), repeats=25)
    oc.check_fused_reductions(dev(), shapes=((1, 2 * 512 * 512, 32), (2, 256 * 256, 128)), repeats=3)
    oc.check_fused_reductions(dev(), shapes=((1, 12, 8196), (2, 10, 20000)), repeats=2)      # two-launch fallback (C > 8192)
    # above the threshold: two launches (no activation: at 11.5 M elements some land within rounding of the LeakyReLU kink)
    oc.check_norm(dev(), instance=False, n=2, c=64, h=300, w=300, act='none')


@pytest.mark.parametrize("nmaps,generated,act,c,ch", [(1, True, 'lrelu', 12, 8), (3, True, 'none', 12, 8),
                                                      (2, False, 'lrelu', 40, 12), (3, True, 'lrelu', 64, 32)])
def test_spade(hip_lib, nmaps, generated, act, c, ch):
    oc.check_spade(dev(), nmaps=nmaps, generated=generated, act=act, c=c, ch=ch)
    if c == 64:
        oc.check_spade(dev(), nmaps=nmaps, generated=generated, act=act, c=c, ch=ch, h=32, w=48)
        oc.check_spade(dev(), nmaps=nmaps, generated=generated, act=act, c=c, ch=20, h=32, w=48, strided=True)
    if generated:
        oc.check_spade(dev(), nmaps=nmaps, generated=True, act=act, c=c, ch=ch, strided=True)


def test_warp_compose(hip_lib):
    oc.check_warp_compose(dev())
    oc.check_warp_compose(dev(), b=2, h=96, w=130)

def test_upsample(hip_lib):
    oc.check_upsample(dev())
    oc.check_upsample(dev(), n=2, c=64, h=16, w=16)


def test_warp_values_grads_and_taps(hip_lib):
    oc.check_warp(dev())
    oc.check_warp(dev(), zero_flow=True)
    oc.check_warp(dev(), b=2, c=3, h=128, w=128, mag=30.0)


def test_warp_tap_indices_bit_exact(hip_lib):
    oc.check_warp_index_image(dev())
    oc.check_warp_index_image(dev(), h=64, w=1024)
    oc.check_warp_index_image(dev(), h=35, w=128)


def test_warp_full_size_zero_flow_round_trip(hip_lib):
    """BASELINE size (512x512): the fp32 grid round trip at zero flow must reproduce ATen's taps exactly."""
    import op_checks
    ops, _ = op_checks.pkg()
    h = w = 512
    flow = torch.zeros(1, 2, h, w)
    _, taps = ops.warp_taps(torch.zeros(1, 1, h, w, device=dev()), flow.to(dev()))
    ref = op_checks.O.resample_taps(flow)
    assert torch.equal(taps.cpu(), ref)
    # 76 of 512 columns land on x-1 at zero flow
    xs = torch.arange(w, dtype=torch.int32)
    assert int((ref[0, 0, :, 0] != xs).sum()) > 0


def test_part_masks(hip_lib):
    oc.check_part_masks(dev())


def test_face_boxes_and_crop(hip_lib):
    oc.check_face_ops(dev())


def test_flownet2_native_operators(hip_lib):
    oc.check_flownet_ops(dev())


def test_adam(hip_lib):
    oc.check_adam(dev(), n=100003)


def test_softmax_channels(hip_lib):
    oc.check_softmax(dev())
    oc.check_softmax(dev(), n=1, c=1024, h=4, w=4)


def test_losses_pack_pool(hip_lib):
    oc.check_losses(dev())


@pytest.mark.gpu
def test_every_gemm_tile(hip_lib):
    """all instantiated forward / weight-gradient tiles (incl. split-K atomics and per-sample weights) on ragged geometries"""
    import tile_checks as tc
    tc.run_all(torch.device('cuda:0'))


@pytest.mark.gpu
def test_avgpool3s2(hip_lib):
    oc.check_avgpool3s2(torch.device('cuda:0'))


@pytest.mark.gpu
@pytest.mark.parametrize('nmaps,generated,c,ch', [(1, True, 32, 16), (3, True, 64, 32), (2, False, 16, 8), (1, True, 12, 8)])
def test_spade_with_folded_upsample(hip_lib, nmaps, generated, c, ch):
    """x at half resolution, read through the nearest x2 index; c = 12 takes the general (non-prepared) path"""
    oc.check_spade(torch.device('cuda:0'), nmaps=nmaps, generated=generated, c=c, ch=ch, h=12, w=10, up=True)


def test_conv_groups(hip_lib):
    oc.check_conv_groups(dev(), big=True)


def test_spade_modulation_fused_with_the_shortcut_convolution(hip_lib):
    """bn_s -> conv_s as one kernel (csrc/spade_conv.hip) == the two launches"""
    oc.check_spade_conv_s(dev())                                                     # level-0 widths, folded up-sampling, two maps
    oc.check_spade_conv_s(dev(), c=128, cout=64, chs=(8,), h=9, w=7, up=False)        # two channel tiles, ragged pixel tile
    oc.check_spade_conv_s(dev(), c=64, cout=64, chs=(8, 8, 4), h=12, w=8, up=True, spectral=False, max_gx=1)   # tile walk, three maps
    oc.check_spade_conv_s(dev(), c=128, cout=32, chs=(36,), h=8, w=8, up=False, grad=False)     # no graph: hs is never written
    oc.check_spade_conv_s(dev(), chs=(16, 8), amp=True)                              # `--amp`: f16 GEMMs, half side output
    oc.check_spade_conv_s(dev(), c=128, cout=64, chs=(40,), h=9, w=7, up=False, grad=False, amp=True)
    oc.check_spade_conv_s(dev(), n=2, c=64, cout=32, chs=(32, 32), h=64, w=96, up=True)
    oc.check_spade_conv_s(dev(), n=1, c=128, cout=64, chs=(64, 64), h=48, w=64, up=True, grad=False)


def test_spade_modulation_fused_with_the_3x3_convolution(hip_lib):
    """actvn(bn_0 / bn_1) -> conv_0 / conv_1 as one kernel (csrc/spade_conv3.hip, round 6) == the two launches == the oracle"""
    oc.check_spade_conv3(dev())                                                       # level-0 conv_0: 64 -> 32, folded up-sampling, two maps
    oc.check_spade_conv3(dev(), cout=64, chs=(8, 8, 4), h=13, w=19, up=False, res=True)             # ragged tiles, three maps, residual
    oc.check_spade_conv3(dev(), cout=32, chs=(36,), h=8, w=16, up=False, grad=False, spectral=False)   # no graph; k % 8 != 0
    oc.check_spade_conv3(dev(), cout=64, chs=(32, 32), h=18, w=34, up=True, act='none', grad=False, res=True)
    oc.check_spade_conv3(dev(), n=2, c=64, cout=32, chs=(32, 32), h=64, w=96, up=True)               # many tiles, both workgroups of a CU
    oc.check_spade_conv3(dev(), n=1, c=64, cout=64, chs=(64, 64), h=48, w=80, up=False, res=True, grad=False)


def test_weighted_sum_of_loss_terms(hip_lib):
    oc.check_weighted_sum(dev())
    oc.check_loss_ticket(dev())


def test_softmax_pooling_as_a_weight_gradient_gemm(hip_lib):
    oc.check_pooled_product(dev())
    oc.check_pooled_product(dev(), b=1, c=32, h=8, w=16, seed=98)
    oc.check_pooled_product(dev(), b=2, c=1024, h=16, w=16, seed=99)


def test_spade_two_site_launch(hip_lib):
    oc.check_spade_pair(dev())
    oc.check_spade_pair(dev(), c=32, chs=(8,), h=9, w=7, up=False)


def test_norm_statistics_from_the_conv_epilogue(hip_lib):
    oc.check_conv_stats(dev())


def test_thin_output_convolutions(hip_lib):
    oc.check_thin_conv(dev())


def test_adaptive_avgpool(hip_lib):
    oc.check_adaptive_avgpool(dev())


def test_ordered_split_k(hip_lib):
    oc.check_ordered_split(dev())


def test_conv_with_the_upsampling_folded_into_the_gather(hip_lib):
    """round 5: nn.Upsample(2) -> conv3x3 (embedding decoders, flow decoder) as ONE launch; the shapes of the 512x512 step included"""
    oc.check_conv_up(dev())
    oc.check_conv_up(dev(), n=1, cin=32, h=9, w=7, cout=32, act='none', stats=1)
    oc.check_conv_up(dev(), n=2, cin=64, h=4, w=4, cout=160, k=3)
    oc.check_conv_up_spectral(dev())
    oc.check_conv_up(dev(), n=1, cin=6, h=5, w=6, cout=8, expect_fold=False)
    oc.check_conv_up(dev(), n=1, cin=16, h=6, w=6, cout=16, amp=True, expect_fold=False)
    oc.check_conv_up(dev(), n=2, cin=64, h=64, w=64, cout=32, stats=1, act='none')               # flow decoder 64 -> 32 at a quarter of its size
    oc.check_conv_up(dev(), n=2, cin=256, h=32, w=32, cout=128, act='none')                     # 256 -> 128 at 64x64: the dominant (LD) tile
    oc.check_conv_up(dev(), n=1, cin=128, h=64, w=48, cout=64, act='none')                      # 128x64 tile
    oc.check_conv_up_spectral(dev(), n=2, cin=64, h=64, w=64, cout=32, with_res=False)          # sub-pixel forward under spectral norm
    oc.check_conv_up(dev(), n=2, cin=128, h=128, w=128, cout=32, act='none')                    # image-embedding decoder 128 -> 32 at 256x256
