"""Network-level CPU logic tests: product modules (HIP kernel sources under the SIMT emulator) vs the CPU oracle,
forward values, post-forward state (spectral-norm vectors, BatchNorm running statistics) and every parameter gradient."""
import torch

import model_checks as mc

DEV = torch.device("cpu")


def test_discriminator_tiny(emu_lib):
    mc.check_discriminator(DEV, mc.tiny_opt(), b=1)


def test_generator_adaptive_spade_tiny(emu_lib):
    mc.check_generator(DEV, mc.tiny_opt(), b=2)


def test_generator_warp_combine_tiny(emu_lib):
    mc.check_generator(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True), b=2)


def test_train_step_pose_warp_combine_tiny(emu_lib):
    """D step + G step (losses, all gradients, flat Adam plumbing) of BASELINE configs[2] flags, tiny width."""
    mc.check_train_step(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True), b=2)


def test_train_step_with_the_subpixel_forward_on_every_up_layer(emu_lib, monkeypatch):
    """the forward of nn.Upsample(2) -> conv3x3 as four 2x2-tap launches per parity class (ops._up_subpixel_forward: the layers of
    >= 8192 source pixels at full size) forced onto every such layer of the tiny network: the whole iteration against the oracle"""
    from importlib import import_module
    ops = import_module('few-shot-vid2vid_amd.ops')
    calls, real = [], ops._up_subpixel_forward
    monkeypatch.setattr(ops, '_up_subpixel_forward', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setenv('FSV_UP_SUBPIXEL_MIN', '1')
    mc.check_train_step(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True), b=2)
    assert len(calls) >= 8, calls          # embedding / flow decoders, both generator passes


def test_generator_with_the_3x3_spade_fusion(emu_lib, monkeypatch):
    """`FSV_SPADE_CONV3=1` (round 6, opt-in): at ngf = 16 the generator has SPADE blocks with 64 input channels - their
    `actvn(bn_0) -> conv_0` (64 -> 32) and `actvn(bn_1) -> conv_1` (64 -> 64, + the shortcut) run as ONE kernel each
    (csrc/spade_conv3.hip on the emulator): the generator's forward and backward pass against the oracle at the usual bars, and the
    fused entry point is what ran."""
    from importlib import import_module
    lib = import_module('few-shot-vid2vid_amd.lib')
    monkeypatch.setenv('FSV_SPADE_CONV3', '1')
    seen, real_call = [], lib.call

    def recording_call(name, *a):
        if name == 'fsv_spade_conv3_fwd':
            seen.append((a[17], a[24]))            # (C, Cout)
        return real_call(name, *a)
    monkeypatch.setattr(lib, 'call', recording_call)
    mc.check_generator(DEV, mc.tiny_opt(ngf=16, nff=4, warp_ref=True, spade_combine=True), b=1)
    assert {(64, 32), (64, 64)} <= set(seen), sorted(set(seen))


def test_train_step_in_the_schedule_bench_py_runs(emu_lib):
    """the discriminator step "on a side stream" (issue order on the emulator), the early generator pass picked up by the
    generator-mode call, the two-piece backward - against the oracle like the plain step"""
    mc.check_train_step(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True), b=2, bench_schedule=True)


def test_train_step_face_tiny(emu_lib):
    """BASELINE configs[0] flavour: fewshot_face, adaptive_spade only, B = 1."""
    mc.check_train_step(DEV, mc.tiny_opt(dataset_mode='fewshot_face', input_nc=1), b=1)


def test_train_step_adaptive_discriminator(emu_lib):
    """--netD_subarch adaptive (discriminator.py:104-209): first discriminator layer with weights generated from the reference
    image (encoder, adaptive average pool, Linear, per-sample stride-2 convolution, InstanceNorm); pinned to the reference itself
    by tests/golden/step_face_adaptive_D.pt"""
    mc.check_train_step(DEV, mc.tiny_opt(dataset_mode='fewshot_face', input_nc=1, netD_subarch='adaptive'), b=2)


def test_train_step_add_raw_output_loss(emu_lib):
    """--add_raw_output_loss (generator.py:195-227): the last n_sc_layers blocks a second time on the label embedding alone (second
    spectral-norm power iteration and BatchNorm running-statistics update of the same modules), the raw image through the GAN and
    feature-matching losses; pinned to the reference itself by tests/golden/step_pose_combine_raw.pt"""
    mc.check_train_step(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, add_raw_output_loss=True), b=2)


def test_temporal_second_frame_tiny(emu_lib):
    """previous-frame flow network (shared with the reference branch), warp and SPADE-combine embedding"""
    mc.check_temporal_step(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True), b=1)


def test_train_step_with_vgg_loss_tiny(emu_lib):
    """G step including the VGG19 perceptual loss (13 conv+ReLU, 4 max-pools, frozen weights) at 32x32"""
    mc.check_train_step(DEV, mc.tiny_opt(dataset_mode='fewshot_face', input_nc=1, no_vgg_loss=False, fineSize=32,
                                         loadSize=32, n_downsample_G=3, n_adaptive_layers=2), b=1)


def test_layout_cache_matches_per_call_prep_tiny(emu_lib):
    """persistent K-major weight layouts + grouped refresh after Adam == per-call re-arrangement"""
    mc.check_layout_cache(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32,
                                           n_downsample_G=3, n_adaptive_layers=2), b=1)


def test_train_step_street_one_hot_tiny(emu_lib):
    """BASELINE configs[4] flavour: fewshot_street, integer class maps one-hot encoded on the way in (label_nc classes),
    adaptive_spade only, 2:1 aspect."""
    mc.check_train_step(DEV, mc.tiny_opt(dataset_mode='fewshot_street', label_nc=7, input_nc=3, aspect_ratio=2.0), b=1)


def test_train_step_with_face_discriminator_tiny(emu_lib):
    """BASELINE configs[3] flavour: --add_face_D (face boxes + crops on the device, 6-channel face discriminator, its
    GAN / feature-matching / L1 / VGG terms) on top of the pose flags, VGG loss on (the reference requires it)."""
    mc.check_train_step(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, add_face_D=True,
                                         no_vgg_loss=False), b=1)


def test_temporal_discriminator_tiny(emu_lib):
    """--lambda_temp > 0: netDT on two stacked frames (D terms DT_real / DT_fake, G terms GT_GAN / GT_GAN_Feat)"""
    mc.check_temporal_step(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, lambda_temp=2.0,
                                            fineSize=32, loadSize=32, n_downsample_G=3, n_adaptive_layers=2), b=2)


def test_train_step_two_reference_images_tiny(emu_lib):
    """--n_shot 2: attention module (key / query encoders, energy over 2 * HW reference positions, softmax, weighted
    sum of the reference features), flow / losses on the attended reference"""
    mc.check_train_step(DEV, mc.tiny_opt(dataset_mode='fewshot_face', input_nc=1, n_shot=2, warp_ref=True), b=2)


def test_train_step_with_teacher_flow_tiny(emu_lib):
    """flow_gt / conf_gt present (training without --no_flow_gt): the masked-L1 flow loss F_Flow against the teacher"""
    mc.check_train_step(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, no_flow_gt=False), b=2,
                        with_flow_gt=True)


def test_flownet2_teacher_reduced_width(emu_lib):
    """FlowNet2 (all five sub-networks, tap-grouped 7x7 / 5x5 convolutions, transposed convolutions, cost volume, warps)
    at 1/8 width on the emulator vs the oracle restatement"""
    mc.check_flownet2(DEV, width_div=8, size=64, tol=1e-4)


def test_flownet_teacher_feeds_the_flow_loss(emu_lib):
    mc.check_flownet_wrapper(DEV)


def test_train_step_with_face_refinement_tiny(emu_lib):
    """--refine_face: second generator on the face crops (decoder fed by the encoding of the coarse face), paste-back with
    bilinear resize + clamp; its gradients and the main generator's through the pasted image"""
    mc.check_train_step(DEV, mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, refine_face=True,
                                         fineSize=128, loadSize=128, n_downsample_G=4, n_adaptive_layers=3), b=1)
