"""Mint the golden fixtures under tests/golden/ from the UNMODIFIED reference (NVlabs/few-shot-vid2vid).

Run in the build container only (needs /root/reference):   python oracle/make_golden.py

What is stored (all small):
  * ref_state_layout.json   - state_dict keys and shapes of the reference's netG / netD for the BASELINE flag sets
                              (checkpoint compatibility contract of the product networks);
  * step_<cfg>.pt           - for a narrow network (ngf = ndf = nff = 8, 64x64): outputs of one reference iteration
                              (train.py:58-62) on seeded synthetic inputs: the generated image / flow / mask / warp,
                              the 6 discriminator and 10 generator losses, and per-parameter gradient norms.
                              Weights are NOT stored: tests/model_checks.fill_state derives every parameter from its
                              state_dict key with numpy's Generator, so the reference, the oracle and the product all
                              see identical weights.
  * step_<cfg>_fullsize.pt  - the benchmarked configurations at full width and full resolution (512x512 pose B = 2, 1024x512
                              street): losses, norms and count sketches of outputs and of every parameter gradient (see FULLSIZE);
  * warp_taps.pt            - integer bilinear tap indices selected by ATen's grid_sample through the reference's
                              `resample` (revealed by the backward scatter pattern), for zero / integer / random flows.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')

CONFIGS = {
    'pose_combine': '--dataset_mode fewshot_pose --aspect_ratio 1 --fineSize 64 --loadSize 64 --adaptive_spade --warp_ref '
                    '--spade_combine --remove_face_labels --no_flow_gt --no_vgg_loss --gpu_ids -1 --ngf 8 --ndf 8 --nff 8 '
                    '--batchSize 2',
    'face': '--dataset_mode fewshot_face --fineSize 64 --loadSize 64 --adaptive_spade --no_flow_gt --no_vgg_loss '
            '--gpu_ids -1 --ngf 8 --ndf 8 --batchSize 2',
    'pose_blend': '--dataset_mode fewshot_pose --aspect_ratio 1 --fineSize 64 --loadSize 64 --adaptive_spade --warp_ref '
                  '--no_flow_gt --no_vgg_loss --gpu_ids -1 --ngf 8 --ndf 8 --nff 8 --batchSize 2',
}
CONFIGS['pose_combine_vgg'] = CONFIGS['pose_combine'].replace(' --no_vgg_loss', '')    # + VGG19 perceptual loss
# BASELINE configs[3] flavour: the face discriminator on top of the pose flags (needs the VGG loss, see loss_collector.py:83)
# 128 x 128 so that the 32 x 32 face crops survive torchvision-style VGG19's five max-pools
CONFIGS['pose_face_d'] = CONFIGS['pose_combine_vgg'].replace('--fineSize 64 --loadSize 64', '--fineSize 128 --loadSize 128') \
    + ' --add_face_D'
# temporal discriminator netDT on two stacked frames (base_model.py:270-276); only used by temporal()
CONFIGS['pose_combine_dt'] = CONFIGS['pose_combine'] + ' --lambda_temp 2'
# two reference images: the attention module of generator.py:291-316 and pick_ref
CONFIGS['face_nshot2'] = CONFIGS['face'] + ' --n_shot 2 --warp_ref'
# teacher flow present (training without --no_flow_gt): flow_gt / conf_gt are synthetic stand-ins fed through data_list
CONFIGS['pose_combine_flowgt'] = CONFIGS['pose_combine'].replace(' --no_flow_gt', '')
# second generator on the face crops (face_refiner.py:24-30); 128 x 128 so that the 32 x 32 face survives four stride-2 layers
CONFIGS['pose_refine_face'] = CONFIGS['pose_combine'].replace('--fineSize 64 --loadSize 64', '--fineSize 128 --loadSize 128') \
    + ' --refine_face --n_downsample_G 4 --n_adaptive_layers 3'
# street: integer class maps, one-hot encoded by encode_label (input_process.py:25-45); default aspect_ratio 2 -> 32 x 64
CONFIGS['street'] = ('--dataset_mode fewshot_street --label_nc 7 --fineSize 64 --loadSize 64 --adaptive_spade --no_flow_gt '
                     '--no_vgg_loss --gpu_ids -1 --ngf 8 --ndf 8 --batchSize 2')
# BASELINE configs[0] / SURVEY 8(d) C1 at FULL width (ngf = ndf = 32 defaults), 128 x 128, batch 1: pins the kernels' full-width
# tile plans (128 ... 1024 channels) to the reference itself, not only to the oracle
CONFIGS['face_fullwidth'] = ('--dataset_mode fewshot_face --fineSize 128 --loadSize 128 --adaptive_spade --no_flow_gt --no_vgg_loss '
                             '--gpu_ids -1 --batchSize 1')
# two-scale discriminator pyramid (scripts/face/train_g8_512.sh): AvgPool2d(3, 2, 1, count_include_pad=False) between the scales
CONFIGS['face_numD2'] = CONFIGS['face'] + ' --num_D 2'
# --add_raw_output_loss (generator.py:195-227): the last n_sc_layers blocks a second time on the label embedding alone, the raw
# image through the GAN / feature-matching losses next to the combined one
CONFIGS['pose_combine_raw'] = CONFIGS['pose_combine'] + ' --add_raw_output_loss'
# --netD_subarch adaptive (discriminator.py:104-209): the first discriminator layer's weights generated from the reference image
CONFIGS['face_adaptive_D'] = CONFIGS['face'] + ' --netD_subarch adaptive'
LAYOUT_CONFIGS = {
    'C3_pose_512': '--dataset_mode fewshot_pose --aspect_ratio 1 --fineSize 512 --loadSize 512 --adaptive_spade --warp_ref '
                   '--spade_combine --remove_face_labels --no_flow_gt --no_vgg_loss --gpu_ids -1',
    'C1_face_128': '--dataset_mode fewshot_face --fineSize 128 --loadSize 128 --adaptive_spade --no_flow_gt --no_vgg_loss '
                   '--gpu_ids -1',
}


def layout():
    res = {}
    for name, flags in LAYOUT_CONFIGS.items():
        opt, model = ref_import.build_model(flags.split())
        res[name] = dict(flags=flags,
                         netG={k: list(v.shape) for k, v in model.netG.state_dict().items()},
                         netD={k: list(v.shape) for k, v in model.netD.state_dict().items()})
        del model
    with open(os.path.join(OUT, 'ref_state_layout.json'), 'w') as f:
        json.dump(res, f)


def step(name, flags):
    import model_checks as mc
    ref_import.install_shims()
    from models.loss_collector import loss_backward
    opt, model = ref_import.build_model(flags.split())
    mc.fill_state(model.netG)
    mc.fill_state(model.netD)
    if model.netDf is not None:
        mc.fill_state(model.netDf)
    if getattr(model, 'refine_face', False):
        mc.fill_state(model.netGf)
    for o in (model.optimizer_G, model.optimizer_D):
        for g in o.param_groups:
            g['lr'] = 0.0
    nl = opt.label_nc if opt.label_nc != 0 else opt.input_nc
    h, w = int(opt.fineSize / opt.aspect_ratio), opt.fineSize
    b = opt.batchSize
    if 'street' in opt.dataset_mode:
        tl, ti, rl, ri = mc.synth_street_inputs(b, h, w, 4242, opt.label_nc)
    else:
        tl, ti, rl, ri = mc.with_n_shot(mc.synth_pose_inputs(b, h, w, 4242, nl), opt.n_shot, b, h, w, 4242, nl)
    flow_gt, conf_gt = [None, None], [None, None]
    if name.endswith('_flowgt'):
        flow_gt[0], conf_gt[0] = mc.synth_flow_gt(b, h, w, 4247)
    data = [tl, ti, flow_gt, conf_gt, rl, ri, None, None, None]
    d_losses = model(data, mode='discriminator')
    d_losses = loss_backward(opt, d_losses, model.optimizer_D, 1)
    gD = {k: float(p.grad.norm()) for k, p in model.netD.named_parameters() if p.grad is not None}
    gDf = {k: float(p.grad.norm()) for k, p in model.netDf.named_parameters() if p.grad is not None} \
        if model.netDf is not None else {}
    # K seeded random projections of every parameter gradient next to its norm (model_checks.sketch): a permuted or
    # sign-flipped gradient of equal norm does not pass them
    skD = {k: mc.sketch(k, p.grad, mc.SKETCH_K_GRAD) for k, p in model.netD.named_parameters() if p.grad is not None}
    skDf = {k: mc.sketch(k, p.grad, mc.SKETCH_K_GRAD) for k, p in model.netDf.named_parameters() if p.grad is not None} \
        if model.netDf is not None else {}
    g_losses, generated, prev = model(data, save_images=True, mode='generator')
    g_losses = loss_backward(opt, g_losses, model.optimizer_G, 0)
    gG = {k: float(p.grad.norm()) for k, p in model.netG.named_parameters() if p.grad is not None}
    skG = {k: mc.sketch(k, p.grad, mc.SKETCH_K_GRAD) for k, p in model.netG.named_parameters() if p.grad is not None}
    if getattr(model, 'refine_face', False):
        gG.update({'netGf.' + k: float(p.grad.norm()) for k, p in model.netGf.named_parameters() if p.grad is not None})
        skG.update({'netGf.' + k: mc.sketch('netGf.' + k, p.grad, mc.SKETCH_K_GRAD) for k, p in model.netGf.named_parameters()
                    if p.grad is not None})
    fake, raw, warped, flow, mask, _ = generated

    def t(x):
        return None if x is None else x.detach().clone()
    torch.save(dict(flags=flags, seed=4242, batch=b, size=w, hw=(h, w),
                    d_losses=[float(x) for x in d_losses], g_losses=[float(x) for x in g_losses],
                    loss_names=model.lossCollector.loss_names,
                    fake=t(fake), raw=t(raw), warp=[t(w) for w in warped], flow=[t(f) for f in flow],
                    mask=[t(m) for m in mask], grad_norm_D=gD, grad_norm_G=gG, grad_norm_Df=gDf,
                    grad_sketch_D=skD, grad_sketch_G=skG, grad_sketch_Df=skDf),
               os.path.join(OUT, 'step_%s.pt' % name))
    print(name, 'D', [round(float(x), 5) for x in d_losses[:2]], 'G', [round(float(x), 5) for x in g_losses])


# ---- the benchmarked configurations at FULL size (BASELINE.json configs[2] / [4]): one iteration of the
# unmodified reference on the seeded inputs the hardware tests use.  Nothing of that size can be committed as tensors (the image is
# 6 MB, the generator's gradient 392 MB), so the fixture keeps losses, norms and count sketches (model_checks.sketch: 256 numbers
# per output tensor, 16 per parameter gradient) - and, per quantity, the distance between the reference's fp32 result and the fp64
# evaluation of the same iteration by the oracle (the reference's own `.float()` casts keep it from running in double): the fp32
# arithmetic's own noise floor on these inputs, which the hardware test adds to its tolerance exactly like the inline fp32 / fp64
# oracle pair it replaces (tests/test_fullsize_gpu.py).
FULLSIZE = {
    'pose_fullsize': dict(flags=LAYOUT_CONFIGS['C3_pose_512'] + ' --batchSize 2', seed=21),
    'street_fullsize': dict(flags='--dataset_mode fewshot_street --label_nc 35 --fineSize 1024 --loadSize 1024 --adaptive_spade '
                                  '--no_flow_gt --no_vgg_loss --gpu_ids -1 --batchSize 1', seed=21),
    'pose_face_d_fullsize': dict(flags=LAYOUT_CONFIGS['C3_pose_512'].replace(' --no_vgg_loss', '') + ' --add_face_D --batchSize 2',
                                 seed=21),
}


def fullsize(name):
    import time
    import model_checks as mc
    from oracle import fsv_oracle as O
    ref_import.install_shims()
    from models.loss_collector import loss_backward
    flags, seed = FULLSIZE[name]['flags'], FULLSIZE[name]['seed']
    t0 = time.time()
    opt, model = ref_import.build_model(flags.split())
    sdG0, sdD0 = mc.fill_state(model.netG), mc.fill_state(model.netD)
    sdDf0 = mc.fill_state(model.netDf) if model.netDf is not None else None
    for o in (model.optimizer_G, model.optimizer_D):
        for g in o.param_groups:
            g['lr'] = 0.0
    nl = opt.label_nc if opt.label_nc != 0 else opt.input_nc
    h, w = int(opt.fineSize / opt.aspect_ratio), opt.fineSize
    b = opt.batchSize
    data4 = mc.synth_street_inputs(b, h, w, seed, opt.label_nc) if 'street' in opt.dataset_mode else mc.synth_pose_inputs(b, h, w, seed, nl)
    tl, ti, rl, ri = data4
    data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
    d_losses = loss_backward(opt, model(data, mode='discriminator'), model.optimizer_D, 1)
    gD = {k: p.grad.detach().clone() for k, p in model.netD.named_parameters() if p.grad is not None}
    gDf = {k: p.grad.detach().clone() for k, p in model.netDf.named_parameters() if p.grad is not None} \
        if model.netDf is not None else {}
    g_losses, generated, _ = model(data, save_images=True, mode='generator')
    g_losses = loss_backward(opt, g_losses, model.optimizer_G, 0)
    gG = {k: p.grad.detach().clone() for k, p in model.netG.named_parameters() if p.grad is not None}
    fake, raw, warped, flow, mask, _ = generated
    outs = {'fake': fake[:, 0]}
    if flow[0] is not None:
        outs.update(flow0=flow[0], mask0=mask[0], warp0=warped[0])
    outs = {k: v.detach().clone() for k, v in outs.items()}
    names = model.lossCollector.loss_names
    d_losses, g_losses = [float(x) for x in d_losses], [float(x) for x in g_losses]
    t_ref = time.time() - t0
    del model
    # ---- the same iteration in fp64 on the oracle: the noise floor of the fp32 evaluation --------------------------------------
    t0 = time.time()
    popt = mc.make_opt(**{k: v for k, v in vars(_product_opt(opt)).items()})
    vw = mc._vgg_weights(popt)
    r64 = O.iteration(sdG0, sdD0, O.cfg_from_opt(popt), data4, torch.float64, vw, sdDf0, [None, None], [None, None], None)
    d64, gD64, g64, gG64, gen64, gDf64 = r64
    t_orc = time.time() - t0

    def pack(grads, grads64):
        out = {}
        for k, g in grads.items():
            g6 = grads64[k].detach().double()
            out[k] = dict(norm=float(g.double().norm()), sketch=mc.sketch(k, g, mc.SKETCH_K_GRAD),
                          noise_l2=float((g.double() - g6).norm()), norm64=float(g6.norm()))
        return out
    o64 = {'fake': gen64['fake']}
    if 'flow0' in outs:
        o64.update(flow0=gen64['flow'][0], mask0=gen64['mask'][0], warp0=gen64['warp'][0])
    out_rec = {k: dict(norm=float(v.double().norm()), absmax=float(v.abs().max()), sketch=mc.sketch(k, v, mc.SKETCH_K_IMAGE),
                       noise_l2=float((v.double() - o64[k].detach().double().reshape(v.shape)).norm()),
                       noise_max=float((v.double() - o64[k].detach().double().reshape(v.shape)).abs().max()))
               for k, v in outs.items()}
    torch.save(dict(flags=flags, seed=seed, batch=b, hw=(h, w), loss_names=names, d_losses=d_losses, g_losses=g_losses,
                    d_losses64=[float(x) for x in d64], g_losses64={k: float(v) for k, v in g64.items()},
                    outputs=out_rec, grad_G=pack(gG, gG64), grad_D=pack(gD, gD64), grad_Df=pack(gDf, gDf64),
                    sketch_k=(mc.SKETCH_K_GRAD, mc.SKETCH_K_IMAGE), torch=torch.__version__,
                    minted='reference fp32 iteration %.0f s, oracle fp64 iteration %.0f s' % (t_ref, t_orc)),
               os.path.join(OUT, 'step_%s.pt' % name))
    worst = max((v['noise_l2'] / max(v['norm64'], 1e-30), k) for k, v in pack(gG, gG64).items() if v['norm64'] > 0)
    print(name, 'D', [round(x, 5) for x in d_losses[:2]], 'G', [round(x, 5) for x in g_losses], '| reference %.0f s, oracle fp64 %.0f s'
          % (t_ref, t_orc), '| image fp32-vs-fp64 rel L2 %.2e' % (out_rec['fake']['noise_l2'] / out_rec['fake']['norm']),
          '| worst gradient noise', worst)


def _product_opt(ref_opt):
    """the reference's parsed options -> the namespace of synth.make_opt (same names; only the keys make_opt knows)"""
    import argparse
    import model_checks as mc
    keys = vars(mc.make_opt())
    return argparse.Namespace(**{k: getattr(ref_opt, k, keys[k]) for k in keys})


def temporal(name, flags):
    """two consecutive frames with the previous-frame branch active (train.py:55-62 with data_prev fed back)"""
    import model_checks as mc
    ref_import.install_shims()
    from models.loss_collector import loss_backward
    opt, model = ref_import.build_model(flags.split())
    mc.fill_state(model.netD)
    model.init_temporal_model()
    mc.fill_state(model.netG)
    if opt.lambda_temp > 0:
        mc.fill_state(model.netDT)
    for o in (model.optimizer_G, model.optimizer_D):
        for g in o.param_groups:
            g['lr'] = 0.0
    frames = [mc.synth_pose_inputs(1, 64, 64, 777 + t, 6) for t in range(2)]
    frames[1] = (frames[1][0], frames[1][1], frames[0][2], frames[0][3])
    prev = [None, None, None]
    for t, (tl, ti, rl, ri) in enumerate(frames):
        data = [tl, ti, [None, None], [None, None], rl, ri] + prev
        d_losses = loss_backward(opt, model(data, mode='discriminator'), model.optimizer_D, 1)
        gDT = {k: float(p.grad.norm()) for k, p in model.netDT.named_parameters() if p.grad is not None} \
            if opt.lambda_temp > 0 else {}
        g_losses, generated, prev = model(data, save_images=True, mode='generator')
        g_losses = loss_backward(opt, g_losses, model.optimizer_G, 0)
    gG = {k: float(p.grad.norm()) for k, p in model.netG.named_parameters() if p.grad is not None}
    fake, raw, warped, flow, mask, _ = generated
    torch.save(dict(flags=flags, seed=777, batch=1, size=64, grad_norm_DT=gDT, d_losses=[float(x) for x in d_losses],
                    g_losses=[float(x) for x in g_losses], loss_names=model.lossCollector.loss_names,
                    fake=fake.detach().clone(), warp=[w.detach().clone() for w in warped],
                    flow=[f.detach().clone() for f in flow], mask=[m.detach().clone() for m in mask], grad_norm_G=gG),
               os.path.join(OUT, 'temporal_%s.pt' % name))
    print('temporal', name, 'D', [round(float(x), 5) for x in d_losses[:2]], 'G', [round(float(x), 5) for x in g_losses])


def inference(name, flags):
    """test.py:27,39-41: model.eval(), three consecutive model.inference() calls (first frame, then two frames with the
    previous-frame branch).  opt.isTrain is switched off after construction so that the test-time weight caching of
    generator.py:370,403-416 is what produces frames 1 and 2."""
    import model_checks as mc
    ref_import.install_shims()
    opt, model = ref_import.build_model(flags.split())
    model.init_temporal_model()
    mc.fill_state(model.netG)
    frames = [mc.synth_pose_inputs(1, 64, 64, 909 + t, 6) for t in range(3)]
    ref_label, ref_image = frames[0][2], frames[0][3]
    # eval-mode networks need meaningful buffers: let the running statistics and the spectral-norm vectors settle with
    # a few training-mode passes, and ship those buffers in the fixture (the weights are reproducible from fill_state)
    model.train()
    for it in range(10):
        model.prevs = None
        for (tl, _, _, _) in frames:
            model.inference(tl, ref_label, ref_image)
    model.prevs = None
    buffers = {k: v.detach().clone() for k, v in model.netG.state_dict().items()
               if k.endswith(('running_mean', 'running_var', 'weight_u', 'weight_v'))}
    model.eval()
    opt.isTrain = False
    fakes = []
    for (tl, _, _, _) in frames:
        fake, raw, warped, flow, mask, _ = model.inference(tl, ref_label, ref_image)
        fakes.append(fake.detach().clone())
    torch.save(dict(flags=flags, seed=909, batch=1, size=64, fakes=fakes, buffers=buffers),
               os.path.join(OUT, 'inference_%s.pt' % name))
    print('inference', name, [round(float(f.abs().mean()), 5) for f in fakes])


def flownet2():
    """FlowNet2 teacher (models/networks/flownet2_pytorch/models.py) on CPU: the reference's network python unmodified, its
    three CUDA extension modules backed by the reference's own kernel templates compiled for the host (oracle/build_ref.py,
    ref_import.install_flownet_shims).  The fixture holds the output flow and the state_dict layout (162.5 M parameters:
    weights come from fill_state)."""
    import model_checks as mc
    net = ref_import.build_flownet2()
    mc.fill_state(net, scale=0.6)
    g = torch.Generator().manual_seed(51)
    size, b = 128, 1
    coarse = torch.rand(b, 3, 2, size // 8, size // 8, generator=g)
    frames = torch.nn.functional.interpolate(coarse.view(b, 6, size // 8, size // 8), size=(size, size), mode='bilinear',
                                             align_corners=True).view(b, 3, 2, size, size)
    with torch.no_grad():
        flow = net(frames)
    torch.save(dict(seed=51, size=size, batch=b, flow=flow.clone(),
                    layout={k: list(v.shape) for k, v in net.state_dict().items()}),
               os.path.join(OUT, 'flownet2.pt'))
    print('flownet2', tuple(flow.shape), float(flow.abs().max()), sum(v.numel() for v in net.state_dict().values()))


def finetune(name, flags, iterations=3):
    """test.py with --finetune (scripts/pose/test.sh): eval-mode model, opt.isTrain False, the first inference() call
    adapts 'fc' / 'conv_img' / 'up' layers and the discriminator on the reference image.  The reference hard-codes 100
    iterations (vid2vid_model.py:219); the fixture limits them by shadowing `range` in that module's namespace - the
    reference source itself is untouched."""
    import builtins
    import random
    import model_checks as mc
    ref_import.install_shims()
    opt, model = ref_import.build_model(flags.split())
    mc.fill_state(model.netG)
    mc.fill_state(model.netD)
    frames = [mc.synth_pose_inputs(1, 64, 64, 1313 + t, 6) for t in range(2)]
    ref_label, ref_image = frames[0][2], frames[0][3]
    model.train()
    with torch.no_grad():
        for it in range(10):              # settle running statistics / spectral-norm vectors of G and D
            for (tl, ti, _, _) in frames:
                data = [tl, ti, [None, None], [None, None], ref_label, ref_image, None, None, None]
                model(data, mode='generator')
    buffers = {}
    for net, tag in ((model.netG, 'G'), (model.netD, 'D')):
        buffers[tag] = {k: v.detach().clone() for k, v in net.state_dict().items()
                        if k.endswith(('running_mean', 'running_var', 'weight_u', 'weight_v'))}
    model.eval()
    opt.isTrain = False
    opt.finetune = True
    model.isTrain = model.lossCollector.isTrain = False
    import models.vid2vid_model as vm
    # only finetune's `range(1, iterations + 1)` (two arguments, starting at 1) is shortened
    vm.range = lambda *a: (builtins.range(1, min(a[1], iterations + 1)) if len(a) == 2 and a[0] == 1
                           else builtins.range(*a))
    random.seed(4321)
    fakes = []
    try:
        for (tl, _, _, _) in frames:
            fake, _, _, _, _, _ = model.inference(tl, ref_label, ref_image)
            fakes.append(fake.detach().clone())
    finally:
        del vm.range
    torch.save(dict(flags=flags, seed=1313, rng_seed=4321, iterations=iterations, batch=1, size=64, fakes=fakes,
                    buffers=buffers, conv_img_weight=model.netG.conv_img.weight.detach().clone(),
                    d_first_weight=model.netD.discriminator_0.model0[0].weight.detach().clone()),
               os.path.join(OUT, 'finetune_%s.pt' % name))
    print('finetune', name, [round(float(f.abs().mean()), 5) for f in fakes])


def warp_taps():
    ref_import.install_shims()
    from models.networks.base_network import resample
    g = torch.Generator().manual_seed(77)
    cases = {}
    for name, (h, w) in dict(w512=(6, 512), w128=(5, 128), w35=(7, 35)).items():
        flow = (torch.rand(1, 2, h, w, generator=g) - 0.5) * 30.0
        flow[:, :, 0] = 0.0                                              # zero flow: fp32 round trip of the grid
        flow[:, :, 1] = torch.randint(-4, 5, (1, 2, w), generator=g).float()   # integer flows
        img = torch.zeros(1, 1, h, w, requires_grad=True)
        out = resample(img, flow)
        taps = torch.full((h, w, 2), -1, dtype=torch.int32)
        for y in range(h):
            for x in range(w):
                gsel = torch.zeros_like(out)
                gsel[0, 0, y, x] = 1.0
                (gi,) = torch.autograd.grad(out, img, gsel, retain_graph=True)
                nz = gi[0, 0].nonzero()
                # north-west tap = smallest touched (row, col); a zero-weight east/south tap is simply absent
                taps[y, x, 0] = int(nz[:, 1].min())
                taps[y, x, 1] = int(nz[:, 0].min())
        cases[name] = dict(flow=flow, taps_xy_min=taps)
    torch.save(cases, os.path.join(OUT, 'warp_taps.pt'))


# ---- `--amp O1`: which operations a half-precision policy casts ---------------------------------------------------------------------
# apex (models/models.py:22-26) is neither vendored nor installable, so the `--amp` arithmetic of the oracle (oracle/np_oracle.py)
# cannot be pinned to it.  What CAN be pinned is its cast list: apex O1 patches torch functions by white / black lists
# (apex/amp/lists/functional_overrides.py: conv*, linear, matmul-family in half; losses, softmax, norms, pointwise transcendental ops
# in fp32) and torch.autocast is that policy's descendant inside torch itself.  This fixture runs the UNMODIFIED reference modules
# under torch.autocast('cpu', dtype=torch.float16) and records, per aten operation, the dtype it ran in, and for every convolution /
# linear / batched product its weight shape - tests/test_golden.py::test_amp_cast_list_against_torch_autocast compares the oracle's
# cast list with it.
AUTOCAST_CONFIGS = {
    'street': ('--dataset_mode fewshot_street --label_nc 35 --fineSize 64 --loadSize 64 --adaptive_spade --no_flow_gt --no_vgg_loss '
               '--gpu_ids -1 --ngf 16 --ndf 16 --batchSize 1 --n_downsample_G 3 --n_adaptive_layers 2'),
    'pose_combine': ('--dataset_mode fewshot_pose --aspect_ratio 1 --fineSize 64 --loadSize 64 --adaptive_spade --warp_ref '
                     '--spade_combine --remove_face_labels --no_flow_gt --no_vgg_loss --gpu_ids -1 --ngf 16 --ndf 16 --nff 16 '
                     '--batchSize 1 --n_downsample_G 3 --n_adaptive_layers 2'),
}


def autocast_ops():
    import model_checks as mc
    ref_import.install_shims()
    res = {}
    for name, flags in AUTOCAST_CONFIGS.items():
        opt, model = ref_import.build_model(flags.split())
        mc.fill_state(model.netG)
        mc.fill_state(model.netD)
        nl = opt.label_nc if opt.label_nc != 0 else opt.input_nc
        h, w = int(opt.fineSize / opt.aspect_ratio), opt.fineSize
        b = opt.batchSize
        if 'street' in opt.dataset_mode:
            tl, ti, rl, ri = mc.synth_street_inputs(b, h, w, 4242, opt.label_nc)
        else:
            tl, ti, rl, ri = mc.synth_pose_inputs(b, h, w, 4242, nl)
        data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
        from oracle.op_census import census_mode
        cen = census_mode()
        with torch.autocast('cpu', dtype=torch.float16):
            with cen:
                model(data, save_images=True, mode='generator')          # G forward, D forward on (real, fake), every loss
        res[name] = dict(flags=flags, torch=torch.__version__, ops=cen.ops,
                         convs=sorted(cen.convs), mms=sorted(cen.mms, key=repr))
        half = sum(1 for c in cen.convs if c[2] == 'float16')
        print(name, 'convolutions:', len(cen.convs), 'in half:', half, '| matrix products:', len(cen.mms),
              '| fp32 ops:', sorted(k for k, v in cen.ops.items() if 'float32' in v))
    with open(os.path.join(OUT, 'autocast_ops.json'), 'w') as f:
        json.dump(res, f, indent=0, sort_keys=True)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:                # mint only the named step cases (keeps the other fixtures byte-identical)
        for n in sys.argv[1:]:
            if n == 'flownet2':
                flownet2()
            elif n == 'autocast_ops':
                autocast_ops()
            elif n.startswith('finetune:'):
                finetune(n[9:], CONFIGS[n[9:]])
            elif n.startswith('inference:'):
                inference(n[10:], CONFIGS[n[10:]])
            elif n.startswith('temporal:'):
                temporal(n[9:], CONFIGS[n[9:]])
            elif n in FULLSIZE:
                fullsize(n)
            else:
                step(n, CONFIGS[n])
        sys.exit(0)
    layout()
    for n, f in CONFIGS.items():
        if n != 'pose_combine_dt':
            step(n, f)
    temporal('pose_combine', CONFIGS['pose_combine'])
    temporal('pose_combine_dt', CONFIGS['pose_combine_dt'])
    inference('pose_combine', CONFIGS['pose_combine'])
    warp_taps()
    flownet2()
    finetune('pose_combine', CONFIGS['pose_combine'])
    autocast_ops()
    for n in FULLSIZE:                   # minutes each (one full-size reference iteration + its fp64 oracle twin)
        fullsize(n)
    print('goldens written to', OUT)
