"""Option namespace and synthetic inputs of the hot path's benchmark / parity workloads.

`make_opt` is the reference's option namespace (options/base_options.py:21-132, train_options.py) with its defaults; the
`synth_*` functions draw the seeded tensors every consumer uses - bench.py, the parity tests, the golden-fixture script
(oracle/make_golden.py) - so that all of them see identical inputs."""
import argparse

import torch


def make_opt(**kw):
    """The reference's option namespace (options/base_options.py:21-132, train_options.py) with its defaults."""
    d = dict(
        ngf=32, ndf=32, nff=32, n_downsample_G=5, n_downsample_F=3, n_blocks_F=6, flow_multiplier=20,
        norm_G='spectralspadesyncbatch', norm_F='spectralsyncbatch', norm_D='spectralinstance',
        conv_ks=3, embed_ks=1, spade_ks=1, netS='encoderdecoder', sc_arch='unet', use_label_ref='mul',
        res_for_ref=False, adaptive_conv=False, adaptive_spade=True, no_adaptive_embed=False, n_adaptive_layers=4,
        n_fc_layers=2, n_frames_G=2, n_frames_per_gpu=1, n_frames_D=2, no_flow_gt=True, spade_combine=False,
        n_sc_layers=2, add_raw_output_loss=False, sep_flow_prev=False, no_sep_warp_embed=False, n_shot=1,
        n_downsample_A=2, warp_ref=False, which_model_netD='multiscale', netD_subarch='n_layers', adaptive_D_layers=1, num_D=1,
        n_layers_D=4, gan_mode='hinge', add_face_D=False, lambda_kld=0.0, lambda_feat=10.0, lambda_temp=0.0,
        lambda_flow=10.0, lambda_mask=10.0, lambda_vgg=10.0, lambda_face=10.0, no_ganFeat_loss=False,
        no_vgg_loss=True, no_TTUR=False, lr=0.0004, beta1=0.5, beta2=0.999, isTrain=True, finetune=False,
        dataset_mode='fewshot_pose', label_nc=0, input_nc=6, output_nc=3, aspect_ratio=1.0, fineSize=64, loadSize=64,
        pose_type='both', remove_face_labels=False, refine_face=False, basic_point_only=False, batchSize=2,
        gpu_ids=[0], distributed=False, amp='O0', niter_single=50,
    )
    for k, v in kw.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return argparse.Namespace(**d)


def synth_pose_inputs(b, h, w, seed=1234, n_label=6):
    """SURVEY.md section 8(d) C3-style synthetic tensors: labels U(-1,1) with a DensePose part-id channel, images U(-1,1)."""
    g = torch.Generator().manual_seed(seed)

    def label():
        l = torch.rand(b, 1, n_label, h, w, generator=g) * 2 - 1
        if n_label >= 3:
            part = torch.round(torch.rand(b, 1, h, w, generator=g) * 24) / 24 * 2 - 1
            bg = torch.ones(b, 1, h, w, dtype=torch.bool)
            bg[:, :, h // 8: h - h // 8, w // 6: w - w // 6] = False
            part[bg] = -1.0
            l[:, :, 2] = part
        return l
    def image():
        # band-limited content (bilinearly up-sampled coarse noise) + a little pixel noise, in [-1, 1]
        coarse = torch.rand(b, 3, max(h // 8, 2), max(w // 8, 2), generator=g) * 2 - 1
        img = torch.nn.functional.interpolate(coarse, size=(h, w), mode='bilinear', align_corners=True)
        img = (img + 0.05 * (torch.rand(b, 3, h, w, generator=g) * 2 - 1)).clamp(-1, 1)
        return img.unsqueeze(1)
    tgt_label, ref_label = label(), label()
    tgt_image, ref_image = image(), image()
    return tgt_label, tgt_image, ref_label, ref_image


def with_n_shot(data, n_shot, b, h, w, seed, n_label):
    """n_shot > 1: n_shot different reference (label, image) pairs per sample, [B, n_shot, C, H, W]"""
    if n_shot <= 1:
        return data
    tl, ti, rl, ri = data
    extra = [synth_pose_inputs(b, h, w, seed + 100 * k, n_label) for k in range(1, n_shot)]
    rl = torch.cat([rl] + [e[2] for e in extra], dim=1)
    ri = torch.cat([ri] + [e[3] for e in extra], dim=1)
    return tl, ti, rl, ri


def synth_street_inputs(b, h, w, seed=1234, n_classes=20):
    """SURVEY.md section 8(d) C5-style tensors: integer class maps (as float, blocky regions) and images U(-1,1)."""
    g = torch.Generator().manual_seed(seed)

    def label():
        coarse = torch.randint(0, n_classes, (b, 1, max(h // 8, 2), max(w // 8, 2)), generator=g).float()
        return torch.nn.functional.interpolate(coarse, size=(h, w), mode='nearest').unsqueeze(1)
    _, ti, _, ri = synth_pose_inputs(b, h, w, seed + 1, 1)
    return label(), ti, label(), ri


def synth_flow_gt(b, h, w, seed):
    """stand-in for the FlowNet2 teacher's output: a smooth flow of a few pixels and a binary confidence map"""
    g = torch.Generator().manual_seed(seed)
    coarse = (torch.rand(b, 2, max(h // 8, 2), max(w // 8, 2), generator=g) - 0.5) * 6
    flow = torch.nn.functional.interpolate(coarse, size=(h, w), mode='bilinear', align_corners=True).unsqueeze(1)
    conf = (torch.rand(b, 1, 1, h, w, generator=g) < 0.7).float()
    return flow, conf
