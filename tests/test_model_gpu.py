"""Network-level parity on a real MI355X (configs of BASELINE.json at reduced width/resolution so the CPU oracle
finishes in seconds; the full-size properties live in test_fullsize_gpu.py)."""
import pytest
import torch

import model_checks as mc

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def test_discriminator(hip_lib):
    mc.check_discriminator(dev(), mc.tiny_opt(ndf=8, fineSize=128, loadSize=128), b=2)


def test_generator_face_like_adaptive_spade(hip_lib):
    # BASELINE configs[0]/[1] flavour: adaptive_spade only, 1-channel labels
    mc.check_generator(dev(), mc.tiny_opt(ngf=8, dataset_mode='fewshot_face', input_nc=1, fineSize=128, loadSize=128), b=2)


def test_generator_pose_warp_combine(hip_lib):
    # BASELINE configs[2] flavour: adaptive_spade + warp_ref + spade_combine
    mc.check_generator(dev(), mc.tiny_opt(ngf=8, nff=8, warp_ref=True, spade_combine=True, fineSize=128, loadSize=128), b=2)


def test_generator_pose_warp_blend(hip_lib):
    mc.check_generator(dev(), mc.tiny_opt(ngf=8, nff=8, warp_ref=True, fineSize=128, loadSize=128), b=2)


def test_train_step_pose_warp_combine(hip_lib):
    mc.check_train_step(dev(), mc.tiny_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True, remove_face_labels=True,
                                           fineSize=128, loadSize=128), b=2)


def test_train_step_face(hip_lib):
    mc.check_train_step(dev(), mc.tiny_opt(ngf=8, ndf=8, dataset_mode='fewshot_face', input_nc=1, fineSize=128, loadSize=128), b=2)


def test_temporal_second_frame(hip_lib):
    mc.check_temporal_step(dev(), mc.tiny_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True, remove_face_labels=True,
                                              fineSize=128, loadSize=128), b=2)


def test_train_step_with_vgg_loss(hip_lib):
    mc.check_train_step(dev(), mc.tiny_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True, remove_face_labels=True,
                                           no_vgg_loss=False, fineSize=64, loadSize=64), b=1)


def test_train_step_street_one_hot(hip_lib):
    """BASELINE configs[4] flavour (fp32): integer class maps -> one-hot, 2:1 aspect"""
    mc.check_train_step(dev(), mc.tiny_opt(ngf=8, ndf=8, dataset_mode='fewshot_street', label_nc=12, input_nc=3,
                                           aspect_ratio=2.0, fineSize=128, loadSize=128), b=2)


def test_train_step_with_face_discriminator(hip_lib):
    """BASELINE configs[3] flavour: --add_face_D on the pose flags (device-side face boxes, crop kernels, netDf)"""
    mc.check_train_step(dev(), mc.tiny_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True, remove_face_labels=True,
                                           add_face_D=True, no_vgg_loss=False, fineSize=128, loadSize=128), b=2)


def test_temporal_discriminator(hip_lib):
    """--lambda_temp > 0: netDT on two stacked frames"""
    mc.check_temporal_step(dev(), mc.tiny_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True,
                                              remove_face_labels=True, lambda_temp=2.0, fineSize=128, loadSize=128), b=2)


def test_train_step_two_reference_images(hip_lib):
    """--n_shot 2: attention over two reference images (two per-sample GEMMs + channel softmax), attended-reference warp"""
    mc.check_train_step(dev(), mc.tiny_opt(ngf=8, ndf=8, nff=8, dataset_mode='fewshot_face', input_nc=1, n_shot=2,
                                           warp_ref=True, fineSize=128, loadSize=128), b=2)


def test_flownet2_teacher_reduced_width(hip_lib):
    mc.check_flownet2(dev(), width_div=4, size=128, b=2, tol=1e-3)


def test_train_step_with_face_refinement(hip_lib):
    """--refine_face: face generator on device-cropped faces, bilinear paste-back"""
    mc.check_train_step(dev(), mc.tiny_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True, remove_face_labels=True,
                                           refine_face=True, fineSize=128, loadSize=128, n_downsample_G=4,
                                           n_adaptive_layers=3), b=2, grad_tol=3e-2)      # hardware record: 1.56e-2 (fc_spade_1_1.0.bias)
