"""CPU logic tests: the HIP kernel sources executed by the SIMT emulator, compared with the CPU oracle.

These do not replace the `gpu` parity tests (tests/test_ops_gpu.py) - they make sure indexing, tiling, bounds
and the maths of every kernel are right before a GPU minute is spent.
"""
import pytest
import torch

import op_checks as oc

DEV = torch.device("cpu")


@pytest.mark.parametrize("cfg", [
    (1, 8, 8, 8, 8, 3, 1, 1, 'lrelu', True),
    (2, 4, 9, 7, 5, 3, 2, 1, 'none', True),
    (1, 6, 10, 10, 12, 3, 1, 1, 'tanh', True),       # Cin % 4 != 0 -> scalar gather path
    (1, 20, 9, 9, 32, 4, 2, 2, 'lrelu', True),        # PatchGAN first layer geometry (k4 s2 p2)
    (1, 8, 7, 7, 16, 4, 1, 2, 'sigmoid', False),      # PatchGAN last layers (k4 s1 p2)
    (1, 64, 6, 6, 130, 1, 1, 0, 'lrelu', True),
    (1, 32, 9, 9, 48, 1, 1, 0, 'none', True),        # small-K weight-gradient tiles (32 / 64 rows)
    (1, 64, 9, 9, 64, 1, 1, 0, 'none', True),
    (2, 32, 5, 7, 130, 1, 1, 0, 'none', False),       # 1x1, Cout not a multiple of 32
])
def test_conv(emu_lib, cfg):
    n, cin, h, w, cout, k, s, p, act, bias = cfg
    oc.check_conv(DEV, n, cin, h, w, cout, k, s, p, act=act, bias=bias)


def test_conv_spectral_residual(emu_lib):
    oc.check_conv_sn_res(DEV)


def test_layout_cache(emu_lib):
    oc.check_layout_cache(DEV)


def test_spectral_power_iteration(emu_lib):
    oc.check_spectral_power_iteration(DEV, shapes=((40, 300), (130, 70), (200, 520)))


def test_split_cols_backward_is_one_concatenation_with_zeros_for_unused_pieces(emu_lib):
    """ops.split_cols == torch.split(dim=1) in values and gradients, including pieces nobody reads (undefined gradients: the
    custom backward fills them from a cached zero block instead of one zero-fill launch each)"""
    ops, conv = oc.pkg()
    g = torch.Generator().manual_seed(3)
    f = torch.randn(3, 40, generator=g)
    sizes = [12, 4, 12, 4, 8]
    a = f.clone().requires_grad_(True)
    b = f.clone().requires_grad_(True)
    pa, pb = ops.split_cols(a, sizes), torch.split(b, sizes, dim=1)
    assert all(torch.equal(x, y) for x, y in zip(pa, pb))
    wts = [torch.randn(3, n, generator=g) for n in sizes]
    use = (0, 1, 3)                                    # pieces 2 and 4 (the tail) are never read
    sum((pa[i] * wts[i]).sum() for i in use).backward()
    sum((pb[i] * wts[i]).sum() for i in use).backward()
    assert torch.equal(a.grad, b.grad)
    # a second call re-uses the cached zeros and must not have been written to
    a2 = f.clone().requires_grad_(True)
    (ops.split_cols(a2, sizes)[4] * wts[4]).sum().backward()
    assert torch.equal(a2.grad[:, :32], torch.zeros(3, 32)) and torch.equal(a2.grad[:, 32:], wts[4])


def test_cat_and_pad_forms(emu_lib):
    oc.check_cat_and_pad(DEV)


def test_deferred_wgrad_finalize(emu_lib):
    oc.check_deferred_wgrad(DEV)


def test_linear(emu_lib):
    oc.check_linear(DEV)


def test_batch_conv(emu_lib):
    oc.check_batch_conv(DEV)


@pytest.mark.parametrize("instance,affine,act", [(False, True, 'lrelu'), (True, True, 'lrelu'), (False, False, 'none')])
def test_norm(emu_lib, instance, affine, act):
    oc.check_norm(DEV, instance=instance, affine=affine, act=act)


@pytest.mark.parametrize("n,c,h,w", [(2, 40, 40, 33), (1, 260, 9, 20), (2, 10, 37, 41)])
def test_norm_multi_chunk_multi_slab(emu_lib, n, c, h, w):
    """pixel counts / channel counts that span several reduction chunks and channel slabs (vector and scalar paths)"""
    oc.check_norm(DEV, instance=False, n=n, c=c, h=h, w=w)
    oc.check_norm(DEV, instance=True, n=n, c=c, h=h, w=w)


def test_fused_reductions(emu_lib):
    oc.check_fused_reductions(DEV, shapes=((1, 4096, 32), (2, 1000, 7), (1, 300, 260), (4, 64, 512)), repeats=2)


def test_reductions_of_large_tensors(emu_lib):
    # 70000 rows of 64 channels, 3 groups x 40000 rows (InstanceNorm form): above the one-launch threshold, many pixel chunks
    oc.check_fused_reductions(DEV, shapes=((1, 70000, 64), (3, 40000, 8)), repeats=1)


def test_fused_reductions_wide_rows_fall_back(emu_lib):
    # C > 8192 on a small tensor needs more channel slabs than there are ticket counters: the one-launch entry points hand
    # over to the two-launch form instead of refusing (round-2 advisor finding)
    oc.check_fused_reductions(DEV, shapes=((1, 12, 8196), (1, 9, 16384), (2, 10, 20000)), repeats=1)


@pytest.mark.parametrize("nmaps,generated,act,c,ch", [(1, True, 'lrelu', 12, 8), (3, True, 'none', 12, 8),
                                                      (2, False, 'lrelu', 40, 12), (3, True, 'lrelu', 32, 20),
                                                      (2, False, 'none', 48, 12), (1, True, 'lrelu', 64, 32)])
def test_spade(emu_lib, nmaps, generated, act, c, ch):
    oc.check_spade(DEV, nmaps=nmaps, generated=generated, act=act, c=c, ch=ch)
    if generated:
        oc.check_spade(DEV, nmaps=nmaps, generated=True, act=act, c=c, ch=ch, strided=True)


def test_warp_compose(emu_lib):
    oc.check_warp_compose(DEV)


def test_upsample(emu_lib):
    oc.check_upsample(DEV)


def test_warp_values_grads_and_taps(emu_lib):
    oc.check_warp(DEV)
    oc.check_warp(DEV, zero_flow=True)


def test_warp_tap_indices_bit_exact(emu_lib):
    oc.check_warp_index_image(DEV)


def test_part_masks(emu_lib):
    oc.check_part_masks(DEV)


def test_face_boxes_and_crop(emu_lib):
    oc.check_face_ops(DEV)


def test_flownet2_native_operators(emu_lib):
    oc.check_flownet_ops(DEV)


def test_adam(emu_lib):
    oc.check_adam(DEV, n=300)


def test_softmax_channels(emu_lib):
    oc.check_softmax(DEV)
    oc.check_softmax(DEV, n=1, c=1024, h=4, w=4)


def test_losses_pack_pool(emu_lib):
    oc.check_losses(DEV)


def test_avgpool3s2(emu_lib):
    oc.check_avgpool3s2(torch.device('cpu'))


@pytest.mark.parametrize('nmaps,generated,c,ch', [(1, True, 32, 16), (3, True, 64, 32), (2, False, 16, 8), (1, True, 12, 8)])
def test_spade_with_folded_upsample(emu_lib, nmaps, generated, c, ch):
    """x at half resolution, read through the nearest x2 index; c = 12 takes the general (non-prepared) path"""
    oc.check_spade(DEV, nmaps=nmaps, generated=generated, c=c, ch=ch, h=12, w=10, up=True)


def test_conv_groups(emu_lib):
    oc.check_conv_groups(DEV)


def test_spade_modulation_fused_with_the_shortcut_convolution(emu_lib):
    """bn_s -> conv_s as one kernel (csrc/spade_conv.hip) == the two launches"""
    oc.check_spade_conv_s(DEV)                                                     # level-0 widths, folded up-sampling, two maps
    oc.check_spade_conv_s(DEV, c=128, cout=64, chs=(8,), h=9, w=7, up=False)        # two channel tiles, ragged pixel tile
    oc.check_spade_conv_s(DEV, c=64, cout=64, chs=(8, 8, 4), h=12, w=8, up=True, spectral=False, max_gx=1)   # tile walk, three maps
    oc.check_spade_conv_s(DEV, c=128, cout=32, chs=(36,), h=8, w=8, up=False, grad=False)     # no graph: hs is never written
    oc.check_spade_conv_s(DEV, chs=(16, 8), amp=True)                              # `--amp`: f16 GEMMs, half side output
    oc.check_spade_conv_s(DEV, c=128, cout=64, chs=(40,), h=9, w=7, up=False, grad=False, amp=True)


def test_spade_modulation_fused_with_the_3x3_convolution(emu_lib):
    """actvn(bn_0 / bn_1) -> conv_0 / conv_1 as one kernel (csrc/spade_conv3.hip, round 6) == the two launches == the oracle"""
    oc.check_spade_conv3(DEV)                                                       # level-0 conv_0: 64 -> 32, folded up-sampling, two maps
    oc.check_spade_conv3(DEV, cout=64, chs=(8, 8, 4), h=13, w=19, up=False, res=True)             # ragged tiles, three maps, residual
    oc.check_spade_conv3(DEV, cout=32, chs=(36,), h=8, w=16, up=False, grad=False, spectral=False)   # no graph: hs never written; k % 8 != 0
    oc.check_spade_conv3(DEV, cout=64, chs=(32, 32), h=18, w=34, up=True, act='none', grad=False, res=True)


def test_weighted_sum_of_loss_terms(emu_lib):
    oc.check_weighted_sum(DEV)
    oc.check_loss_ticket(DEV)


def test_softmax_pooling_as_a_weight_gradient_gemm(emu_lib):
    oc.check_pooled_product(DEV)
    oc.check_pooled_product(DEV, b=1, c=32, h=8, w=16, seed=98)


def test_spade_two_site_launch(emu_lib):
    oc.check_spade_pair(DEV)
    oc.check_spade_pair(DEV, c=32, chs=(8,), h=9, w=7, up=False)


def test_norm_statistics_from_the_conv_epilogue(emu_lib):
    oc.check_conv_stats(DEV)


def test_thin_output_convolutions(emu_lib):
    oc.check_thin_conv(DEV)


def test_adaptive_avgpool(emu_lib):
    oc.check_adaptive_avgpool(DEV)


def test_ordered_split_k(emu_lib):
    oc.check_ordered_split(DEV)


def test_conv_with_the_upsampling_folded_into_the_gather(emu_lib):
    """round 5: nn.Upsample(2) -> conv3x3 of the label / image embedding decoders and the flow network's decoder as ONE launch"""
    oc.check_conv_up(DEV)                                                           # 64x64-tile class, LeakyReLU epilogue
    oc.check_conv_up(DEV, n=1, cin=32, h=9, w=7, cout=32, act='none', stats=1)       # odd source size, BatchNorm statistics ride along
    oc.check_conv_up(DEV, n=2, cin=64, h=4, w=4, cout=160, k=3)
    oc.check_conv_up_spectral(DEV)                      # wide output: the 8-wave tiles
    oc.check_conv_up(DEV, n=1, cin=8, h=5, w=6, cout=8, k=1)                         # 1x1
    oc.check_conv_up(DEV, n=1, cin=6, h=5, w=6, cout=8, expect_fold=False)           # scalar-gather channels: materialised
    oc.check_conv_up(DEV, n=1, cin=16, h=5, w=6, cout=3, expect_fold=False)          # thin head: materialised
    oc.check_conv_up(DEV, n=1, cin=16, h=6, w=6, cout=16, amp=True, expect_fold=False)   # half-precision path: materialised
    oc.check_conv_up(DEV, n=2, cin=8, h=64, w=64, cout=8, act='none')                # >= 8192 source pixels: the sub-pixel forward
    from fsv2v_amd.layout_cache import LayoutCache
    oc.check_conv_up(DEV, n=2, cin=8, h=64, w=64, cout=40, act='none', cache=LayoutCache())   # ... from the cached summed-tap layouts
    oc.check_conv_up(DEV, n=1, cin=20, h=6, w=7, cout=24, cache=LayoutCache())       # cin % 8 != 0: data gradient cached, forward per call
    oc.check_conv_up_spectral(DEV, n=2, cin=8, h=64, w=64, cout=8, with_res=False)   # ... of a spectral-normalised layer
