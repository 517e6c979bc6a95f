"""Golden fixtures minted from the unmodified reference (oracle/make_golden.py) pin

  * the CPU oracle (any box, `not gpu`): same losses / images / gradient norms / warp taps as the reference;
  * the product networks' checkpoint layout (state_dict keys and shapes identical to the reference's);
  * the product on a real MI355X (`gpu`): one reference iteration reproduced within 1e-3 (outputs, losses).
"""
import json
import os

import pytest
import torch

import model_checks as mc
from oracle import fsv_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['pose_combine', 'face', 'pose_blend', 'pose_combine_vgg', 'street', 'pose_face_d', 'face_nshot2',
         'pose_combine_flowgt', 'pose_refine_face', 'face_fullwidth', 'face_numD2', 'pose_combine_raw', 'face_adaptive_D']      # face_fullwidth: ngf = ndf = 32, 128 x 128, B = 1 (C1)


def _opt_from_flags(flags):
    """the reference's command line -> the option namespace used by the product / oracle (defaults identical)"""
    toks = flags.split()
    kw = {}
    ints = {'--ngf': 'ngf', '--ndf': 'ndf', '--nff': 'nff', '--fineSize': 'fineSize', '--loadSize': 'loadSize',
            '--batchSize': 'batchSize', '--n_downsample_G': 'n_downsample_G', '--n_adaptive_layers': 'n_adaptive_layers',
            '--num_D': 'num_D'}
    i = 0
    while i < len(toks):
        t = toks[i]
        if t in ints:
            kw[ints[t]] = int(toks[i + 1]); i += 2
        elif t == '--dataset_mode':
            kw['dataset_mode'] = toks[i + 1]; i += 2
        elif t == '--label_nc':
            kw['label_nc'] = int(toks[i + 1]); i += 2
        elif t == '--lambda_temp':
            kw['lambda_temp'] = float(toks[i + 1]); i += 2
        elif t == '--n_shot':
            kw['n_shot'] = int(toks[i + 1]); i += 2
        elif t == '--netD_subarch':
            kw['netD_subarch'] = toks[i + 1]; i += 2
        elif t == '--aspect_ratio':
            kw['aspect_ratio'] = float(toks[i + 1]); i += 2
        elif t == '--gpu_ids':
            i += 2
        elif t in ('--adaptive_spade', '--warp_ref', '--spade_combine', '--remove_face_labels', '--no_flow_gt',
                   '--no_vgg_loss', '--add_face_D', '--refine_face', '--add_raw_output_loss'):
            kw[t[2:]] = True; i += 1
        else:
            raise ValueError(t)
    if kw.get('dataset_mode') == 'fewshot_face':
        kw.setdefault('input_nc', 1)
    if kw.get('dataset_mode') == 'fewshot_street':          # data/fewshot_street_dataset.py:18-22 defaults
        kw.setdefault('input_nc', 3)
        kw.setdefault('aspect_ratio', 2.0)
    kw.setdefault('no_vgg_loss', False)          # the reference's default: VGG perceptual loss on
    return mc.make_opt(**kw)


def _inputs(g, opt):
    """the seeded synthetic tensors the fixture was minted on"""
    if 'street' in opt.dataset_mode:
        h, w = g['hw']
        return mc.synth_street_inputs(g['batch'], h, w, g['seed'], opt.label_nc)
    nl = opt.label_nc if opt.label_nc != 0 else opt.input_nc
    return mc.with_n_shot(mc.synth_pose_inputs(g['batch'], g['size'], g['size'], g['seed'], nl), opt.n_shot, g['batch'],
                          g['size'], g['size'], g['seed'], nl)


def _flow_gt(case, g):
    if not case.endswith('_flowgt'):
        return [None, None], [None, None]
    flow, conf = mc.synth_flow_gt(g['batch'], g['size'], g['size'], g['seed'] + 5)
    return [flow, None], [conf, None]


def _load(case):
    return torch.load(os.path.join(GOLD, 'step_%s.pt' % case), weights_only=False)


def _t(x):
    """reference outputs are [B, T, ...] (fake, raw) or already 4-D (warped, flow, mask)"""
    return x[:, 0] if x.dim() == 5 else x


def _rel(a, b):
    a, b = _t(a), _t(b)
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


@pytest.mark.parametrize('case', CASES)
def test_oracle_reproduces_reference_iteration(case):
    g = _load(case)
    opt = _opt_from_flags(g['flags'])
    net = mc._net()
    M = mc._model()
    model = M.create_model(opt)            # used only for parameter shapes / names (CPU, nothing is launched)
    sdG0, sdD0 = mc.fill_state(model.netG), mc.fill_state(model.netD)
    data = _inputs(g, opt)
    sdDf0 = mc.fill_state(model.netDf) if model.netDf is not None else None
    sdGf0 = mc.fill_state(model.netGf) if model.netGf is not None else None
    d_losses, gD, g_losses, gG, gen, gDf = mc._oracle_iteration(sdG0, sdD0, O.cfg_from_opt(opt), data, torch.float32,
                                                           mc._vgg_weights(opt), sdDf0, *_flow_gt(case, g), sdGf0=sdGf0)
    names = g['loss_names']
    for i in range(2, len(d_losses)):              # Df_real, Df_fake with --add_face_D
        assert abs(float(d_losses[i]) - g['d_losses'][i]) <= 1e-5 * max(1.0, abs(g['d_losses'][i])), i
    for k, ref in g.get('grad_norm_Df', {}).items():
        assert abs(float(gDf[k].norm()) - ref) <= 1e-4 * max(ref, 1e-6), k
    assert abs(float(d_losses[0]) - g['d_losses'][0]) <= 1e-5 * max(1.0, abs(g['d_losses'][0]))
    assert abs(float(d_losses[1]) - g['d_losses'][1]) <= 1e-5 * max(1.0, abs(g['d_losses'][1]))
    for k, v in g_losses.items():
        ref = g['g_losses'][names.index(k)]
        assert abs(float(v) - ref) <= 1e-5 * max(1.0, abs(ref)), (k, float(v), ref)
    assert _rel(gen['fake'], g['fake']) <= 1e-5
    if g['flow'][0] is not None:
        assert _rel(gen['flow'][0], g['flow'][0]) <= 1e-5
        assert _rel(gen['mask'][0], g['mask'][0]) <= 1e-5
        assert _rel(gen['warp'][0], g['warp'][0]) <= 1e-5
    for k, ref in g['grad_norm_D'].items():
        assert abs(float(gD[k].norm()) - ref) <= 1e-4 * max(ref, 1e-6), k
    med = sorted(g['grad_norm_G'].values())[len(g['grad_norm_G']) // 2]
    for k, ref in g['grad_norm_G'].items():
        assert abs(float(gG[k].norm()) - ref) <= 1e-3 * max(ref, 1e-2 * med), k
    # ... and, element-wise through 16 seeded projections per parameter (model_checks.sketch: the sketch distance estimates the L2
    # distance of the two gradients; a permuted / sign-flipped gradient of equal norm sits at ~1.4 x its norm)
    for grads, sk, norms in ((gD, g['grad_sketch_D'], g['grad_norm_D']), (gG, g['grad_sketch_G'], g['grad_norm_G']),
                             (gDf, g.get('grad_sketch_Df', {}), g.get('grad_norm_Df', {}))):
        assert set(sk) == set(norms)
        top = max(norms.values()) if norms else 0.0
        for k, ref in sk.items():
            if norms[k] < 1e-4 * top:      # mathematically zero (a conv bias in front of a normalisation): rounding noise on both sides
                continue
            dist = mc.sketch_distance(mc.sketch(k, grads[k], mc.SKETCH_K_GRAD), ref)
            assert dist <= 1e-3 * max(norms[k], 1e-2 * med), (k, dist, norms[k])


def test_oracle_warp_taps_match_aten_selection():
    cases = torch.load(os.path.join(GOLD, 'warp_taps.pt'), weights_only=False)
    for name, c in cases.items():
        taps = O.resample_taps(c['flow'])[0]
        gold = c['taps_xy_min']
        h, w = gold.shape[:2]
        # ATen's backward touched (y_n, x_w) unless the whole north / west weight was exactly 0 (integer coordinate),
        # in which case the smallest touched index is the tap itself
        assert torch.equal(taps[..., 0], gold[..., 0]) or bool(((taps[..., 0] == gold[..., 0]) |
                                                                 (taps[..., 0] + 1 == gold[..., 0])).all()), name
        assert bool(((taps[..., 1] == gold[..., 1]) | (taps[..., 1] + 1 == gold[..., 1])).all()), name
        frac_exact = float((taps[..., 0] == gold[..., 0]).float().mean())
        assert frac_exact > 0.95, (name, frac_exact)


@pytest.mark.parametrize('cfg', ['C3_pose_512', 'C1_face_128'])
def test_product_state_dict_layout_equals_reference(cfg):
    layout = json.load(open(os.path.join(GOLD, 'ref_state_layout.json')))[cfg]
    opt = _opt_from_flags(layout['flags'])
    M = mc._model()
    with torch.device('meta'):
        model = M.create_model(opt)
    for net, ref in ((model.netG, layout['netG']), (model.netD, layout['netD'])):
        mine = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert set(mine) == set(ref), (sorted(set(mine) ^ set(ref))[:10])
        bad = [k for k in ref if mine[k] != ref[k]]
        assert not bad, bad[:10]


def _check_grad_sketches(net, ref_sketches, ref_norms, tag, tol=3e-2, prefix=''):
    """element-wise twin of _check_grad_norms: the distance between the product's and the reference's count sketch estimates the L2
    distance of the two gradients (spread ~ sqrt(2 / 16)): held to 2e-2 (the step-level gradient band of model_checks.check_train_step
    for these narrow networks: activations within rounding of a LeakyReLU / hinge kink move single entries by O(1)) x 1.5 for the
    estimator's spread; twice that for the flow network (the warp is piecewise smooth in the flow)."""
    if not ref_sketches:
        return
    med = max(sorted(ref_norms.values())[len(ref_norms) // 2], 1e-2 * max(ref_norms.values()))
    for name, prm in net.named_parameters():
        if name not in ref_sketches:
            continue
        dist = mc.sketch_distance(mc.sketch(prefix + name, prm.grad, mc.SKETCH_K_GRAD), ref_sketches[name])
        t = tol * 2 if 'flow_network' in name else tol
        assert dist <= t * max(ref_norms[name], 1e-2 * med), (tag, name, dist, ref_norms[name])


def _check_grad_norms(net, ref_norms, tag, tol=1e-2):
    """per-parameter gradient norms of the reference iteration (lr = 0, so .grad survives the optimiser step).  The band
    is relative to max(norm, 1 % of the network's median norm, 1e-4 of its largest norm): gradients that are mathematically
    zero (a conv bias in front of a normalisation) are rounding noise on both sides."""
    if not ref_norms:
        return
    med = max(sorted(ref_norms.values())[len(ref_norms) // 2], 1e-2 * max(ref_norms.values()))
    seen = 0
    for name, prm in net.named_parameters():
        if name not in ref_norms:
            continue
        assert prm.grad is not None, (tag, name)
        got, ref = float(prm.grad.norm()), ref_norms[name]
        assert abs(got - ref) <= tol * max(ref, 1e-2 * med), (tag, name, got, ref)
        seen += 1
    assert seen == len(ref_norms), (tag, seen, len(ref_norms))


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
def test_product_reproduces_reference_iteration_on_gpu(hip_lib, case):
    g = _load(case)
    opt = _opt_from_flags(g['flags'])
    dev = torch.device('cuda:0')
    M = mc._model()
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    if model.netDf is not None:
        mc.fill_state(model.netDf)
    if model.netGf is not None:
        mc.fill_state(model.netGf)
    model = model.to(dev).train()
    opt_G, opt_D = model.build_optimizers()
    opt_G.set_lr(0.0); opt_D.set_lr(0.0)
    nl = opt.label_nc if opt.label_nc != 0 else opt.input_nc
    tl, ti, rl, ri = [t.to(dev) for t in _inputs(g, opt)]
    fgt, cgt = _flow_gt(case, g)
    data = [tl, ti, [None if t is None else t.to(dev) for t in fgt], [None if t is None else t.to(dev) for t in cgt], rl,
            ri, None, None, None]
    d = M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
    _check_grad_norms(model.netD, g['grad_norm_D'], 'netD')
    _check_grad_sketches(model.netD, g['grad_sketch_D'], g['grad_norm_D'], 'netD')
    if model.netDf is not None:
        _check_grad_norms(model.netDf, g.get('grad_norm_Df', {}), 'netDf')
        _check_grad_sketches(model.netDf, g.get('grad_sketch_Df', {}), g.get('grad_norm_Df', {}), 'netDf')
    gl, generated, _ = model(data, save_images=True, mode='generator')
    gl = M.loss_backward(opt, gl, opt_G, 0)
    nG = {k: v for k, v in g['grad_norm_G'].items() if not k.startswith('netGf.')}
    _check_grad_norms(model.netG, nG, 'netG')
    _check_grad_sketches(model.netG, {k: v for k, v in g['grad_sketch_G'].items() if not k.startswith('netGf.')}, nG, 'netG')
    if model.netGf is not None:
        nGf = {k[6:]: v for k, v in g['grad_norm_G'].items() if k.startswith('netGf.')}
        _check_grad_norms(model.netGf, nGf, 'netGf')
        _check_grad_sketches(model.netGf, {k[6:]: v for k, v in g['grad_sketch_G'].items() if k.startswith('netGf.')}, nGf, 'netGf',
                             prefix='netGf.')
    for i in range(len(d)):
        assert abs(float(d[i]) - g['d_losses'][i]) <= 1e-3 * max(1.0, abs(g['d_losses'][i])), i
    for i, ref in enumerate(g['g_losses']):
        assert abs(float(gl[i]) - ref) <= 1e-3 * max(1.0, abs(ref)), (g['loss_names'][i], float(gl[i]), ref)
    assert _rel(generated[0].cpu(), g['fake']) <= 1e-3
    if g['flow'][0] is not None:
        assert _rel(generated[3][0].cpu(), g['flow'][0]) <= 1e-3
        assert _rel(generated[4][0].cpu(), g['mask'][0]) <= 1e-3
        assert _rel(generated[2][0].cpu(), g['warp'][0]) <= 1e-3


@pytest.mark.parametrize('case', ['pose_combine', 'pose_combine_dt'])
def test_oracle_reproduces_reference_second_frame(case):
    """previous-frame branch (init_temporal_model): two reference frames, oracle compared on the second one; the _dt
    case adds the temporal discriminator (--lambda_temp 2: DT_real / DT_fake, GT_GAN / GT_GAN_Feat)"""
    g = torch.load(os.path.join(GOLD, 'temporal_%s.pt' % case), weights_only=False)
    opt = _opt_from_flags(g['flags'])
    M = mc._model()
    model = M.create_model(opt)
    mc.fill_state(model.netD)
    model.init_temporal_model()
    mc.fill_state(model.netG)
    sdDT0 = None
    if opt.lambda_temp > 0:
        mc.fill_state(model.netDT)
        sdDT0 = {k: v.detach().clone() for k, v in model.netDT.state_dict().items()}
    sdG0 = {k: v.detach().clone() for k, v in model.netG.state_dict().items()}
    sdD0 = {k: v.detach().clone() for k, v in model.netD.state_dict().items()}
    frames = [mc.synth_pose_inputs(g['batch'], g['size'], g['size'], g['seed'] + t, 6) for t in range(2)]
    frames[1] = (frames[1][0], frames[1][1], frames[0][2], frames[0][3])
    d_losses, gD, g_losses, gG, gen, gDT = mc._oracle_two_frames(sdG0, sdD0, O.cfg_from_opt(opt), frames, torch.float32,
                                                                 sdDT0)
    names = g['loss_names']
    for i in range(len(d_losses)):
        assert abs(float(d_losses[i]) - g['d_losses'][i]) <= 1e-5 * max(1.0, abs(g['d_losses'][i])), i
    if opt.lambda_temp > 0:
        assert len(d_losses) == 6 and 'GT_GAN' in g_losses and g['g_losses'][names.index('GT_GAN_Feat')] > 0
    for k, ref in g.get('grad_norm_DT', {}).items():
        assert abs(float(gDT[k].norm()) - ref) <= 1e-4 * max(ref, 1e-6), k
    for k, v in g_losses.items():
        ref = g['g_losses'][names.index(k)]
        assert abs(float(v) - ref) <= 1e-5 * max(1.0, abs(ref)), (k, float(v), ref)
    assert _rel(gen['fake'], g['fake']) <= 1e-5
    assert _rel(gen['warp'][1], g['warp'][1]) <= 1e-5 and _rel(gen['flow'][1], g['flow'][1]) <= 1e-5
    med = sorted(g['grad_norm_G'].values())[len(g['grad_norm_G']) // 2]
    for k, ref in g['grad_norm_G'].items():
        if k.startswith('flow_network_temp.'):
            k2 = k.replace('flow_network_temp.', 'flow_network_ref.')      # one shared module, two names
        else:
            k2 = k
        # (parameters whose true gradient is zero - conv biases in front of a BatchNorm - hold rounding noise on both
        # sides: the floor keeps them out of the relative comparison)
        assert abs(float(gG[k2].norm()) - ref) <= 1e-3 * max(ref, 5e-2 * med), k


def _inference_setup(g, device=None):
    """product model with the fixture's state: fill_state weights + the settled eval-mode buffers of the reference"""
    opt = _opt_from_flags(g['flags'])
    M = mc._model()
    model = M.create_model(opt)
    model.netG.init_temporal_network()
    mc.fill_state(model.netG)
    sd = model.netG.state_dict()
    for k, v in g['buffers'].items():
        sd[k].copy_(v)
    frames = [mc.synth_pose_inputs(g['batch'], g['size'], g['size'], g['seed'] + t, 6) for t in range(3)]
    return opt, model, frames


def test_oracle_reproduces_reference_inference():
    """test.py path: model.eval(), three Vid2VidModel.inference() frames (first frame, then previous-frame warping)"""
    g = torch.load(os.path.join(GOLD, 'inference_pose_combine.pt'), weights_only=False)
    opt, model, frames = _inference_setup(g)
    sdG = {k: v.detach().clone() for k, v in model.netG.state_dict().items()}
    outs = O.inference_frames(sdG, O.cfg_from_opt(opt), [f[0] for f in frames], frames[0][2], frames[0][3],
                              warp_prev=True, n_frames_G=opt.n_frames_G)
    for got, ref in zip(outs, g['fakes']):
        assert _rel(got, ref) <= 1e-5


def _check_product_inference(dev):
    g = torch.load(os.path.join(GOLD, 'inference_pose_combine.pt'), weights_only=False)
    opt, model, frames = _inference_setup(g)
    model = model.to(dev).eval()
    opt.isTrain = False                    # test-time weight caching (generator.py:370,403-416)
    ref_label, ref_image = frames[0][2].to(dev), frames[0][3].to(dev)
    for t, (f, ref) in enumerate(zip(frames, g['fakes'])):
        data = [f[0].to(dev), None, None, None, ref_label, ref_image, None, None, None]
        fake = model(data)[0]              # default mode = inference, like test.py:40
        assert _rel(fake.cpu(), ref) <= 1e-3, t
    assert model.t == 2 and model.netG._cached_weights is not None


def test_product_inference_emu(emu_lib):
    _check_product_inference(torch.device('cpu'))


@pytest.mark.gpu
def test_product_inference_on_gpu(hip_lib):
    _check_product_inference(torch.device('cuda:0'))


def _flownet2_frames(g):
    gen = torch.Generator().manual_seed(g['seed'])
    size, b = g['size'], g['batch']
    coarse = torch.rand(b, 3, 2, size // 8, size // 8, generator=gen)
    return torch.nn.functional.interpolate(coarse.view(b, 6, size // 8, size // 8), size=(size, size), mode='bilinear',
                                           align_corners=True).view(b, 3, 2, size, size)


def test_flownet2_oracle_and_layout_match_reference():
    """FlowNet2 teacher: the reference's network python (its CUDA extensions replaced by oracle/flownet_oracle.py) vs the
    functional restatement, and the product module's state_dict layout vs the reference's (checkpoint compatibility)."""
    from oracle import flownet_oracle as FO
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    fn = import_module('few-shot-vid2vid_amd.flownet2')
    g = torch.load(os.path.join(GOLD, 'flownet2.pt'), weights_only=False)
    with torch.device('meta'):
        net = fn.FlowNet2()
    mine = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert mine == g['layout'], sorted(set(mine) ^ set(g['layout']))[:10]
    sd = {k: mc.fill_value(k, shape, 0.6) for k, shape in g['layout'].items()}
    with torch.no_grad():
        flow = FO.flownet2(sd, _flownet2_frames(g))
    assert _rel(flow, g['flow']) <= 1e-5


@pytest.mark.gpu
def test_product_flownet2_on_gpu(hip_lib):
    """full-width FlowNet2 (162.5 M parameters) on the HIP kernels vs the reference fixture"""
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    fn = import_module('few-shot-vid2vid_amd.flownet2')
    g = torch.load(os.path.join(GOLD, 'flownet2.pt'), weights_only=False)
    with torch.device('meta'):
        net = fn.FlowNet2()
    net = net.to_empty(device='cuda:0')
    net.load_state_dict({k: mc.fill_value(k, shape, 0.6) for k, shape in g['layout'].items()})
    flow = net(_flownet2_frames(g).to('cuda:0')).cpu()
    assert _rel(flow, g['flow']) <= 1e-3


def _finetune_setup(g):
    import argparse
    opt = _opt_from_flags(g['flags'])
    opt = argparse.Namespace(**{**vars(opt), 'finetune': True, 'finetune_iterations': g['iterations']})
    M = mc._model()
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    for net, tag in ((model.netG, 'G'), (model.netD, 'D')):
        sd = net.state_dict()
        for k, v in g['buffers'][tag].items():
            sd[k].copy_(v)
    frames = [mc.synth_pose_inputs(g['batch'], g['size'], g['size'], g['seed'] + t, 6) for t in range(2)]
    return opt, model, frames


def test_oracle_reproduces_reference_finetune():
    """test.py --finetune (scripts/pose/test.sh): three adaptation iterations (G step then D step on rolled / flipped
    reference images, Python `random` stream), then two inference frames"""
    import random
    g = torch.load(os.path.join(GOLD, 'finetune_pose_combine.pt'), weights_only=False)
    opt, model, frames = _finetune_setup(g)
    sdG = mc._leafify({k: v.detach().clone() for k, v in model.netG.state_dict().items()}, torch.float32)
    sdD = mc._leafify({k: v.detach().clone() for k, v in model.netD.state_dict().items()}, torch.float32)
    cfg = O.cfg_from_opt(opt)
    cfg.isTrain = False
    random.seed(g['rng_seed'])
    O.finetune(sdG, sdD, cfg, frames[0][2], frames[0][3], iterations=g['iterations'])
    assert _rel(sdG['conv_img.weight'].detach(), g['conv_img_weight']) <= 1e-5
    assert _rel(sdD['discriminator_0.model0.0.weight'].detach(), g['d_first_weight']) <= 1e-5
    with torch.no_grad():
        outs = O.inference_frames({k: v.detach() for k, v in sdG.items()}, cfg, [f[0] for f in frames], frames[0][2],
                                  frames[0][3], n_frames_G=opt.n_frames_G)
    for got, ref in zip(outs, g['fakes']):
        assert _rel(got, ref) <= 1e-5


def _check_product_finetune(dev):
    import random
    g = torch.load(os.path.join(GOLD, 'finetune_pose_combine.pt'), weights_only=False)
    opt, model, frames = _finetune_setup(g)
    model = model.to(dev).eval()
    opt.isTrain = False
    model.isTrain = False
    random.seed(g['rng_seed'])
    ref_label, ref_image = frames[0][2].to(dev), frames[0][3].to(dev)
    for t, (f, ref) in enumerate(zip(frames, g['fakes'])):
        fake = model([f[0].to(dev), None, None, None, ref_label, ref_image, None, None, None])[0]
        assert _rel(fake.cpu(), ref) <= 1e-3, t
    assert _rel(model.netG.conv_img.weight.detach().cpu(), g['conv_img_weight']) <= 1e-3
    # Adam turns rounding-level gradients into +-lr steps, so individual discriminator weights can differ by lr x
    # iterations; the tensor as a whole is compared
    dw = model.netD.discriminator_0.model0[0].weight.detach().cpu()
    assert float((dw - g['d_first_weight']).norm() / g['d_first_weight'].norm()) <= 2e-3


def test_product_finetune_emu(emu_lib):
    _check_product_finetune(torch.device('cpu'))


@pytest.mark.gpu
def test_product_finetune_on_gpu(hip_lib):
    _check_product_finetune(torch.device('cuda:0'))


# ------------------------------------------------------------------------------------------------ `--amp O1`: the cast list
@pytest.mark.parametrize('name', ['street', 'pose_combine'])
def test_amp_cast_list_against_torch_autocast(name):
    """*Parity unpinned vs apex* (not vendored, not installable, no CPU path) - but the CAST LIST of the `--amp O1` definition
    (oracle/np_oracle.amp_conv2d installed into oracle/fsv_oracle.py) is pinned here against torch's own half-precision cast policy:
    tests/golden/autocast_ops.json records, for the UNMODIFIED reference modules run under torch.autocast('cpu', float16)
    (oracle/make_golden.py autocast_ops), which aten operations ran in half and which in fp32.  apex O1 patches torch functions by
    the same kind of white / black lists (convolutions, linear and the matmul family in half; losses, norms of vectors, pooling,
    grid_sample in fp32) - torch.autocast is that policy inside torch.

    What is asserted: (1) the oracle executes the same contraction layers as the reference (multiset of convolution weight shapes
    and of matrix products); (2) every convolution the definition runs on half operands is one autocast runs in half - the
    definition never narrows what the policy keeps in fp32; (3) what autocast runs in half and the definition keeps in fp32 is
    exactly the documented set: convolutions the half kernels' contract does not cover (output channels not a multiple of 8 -
    image / flow / mask heads, the discriminators' one-channel output -, per-sample products whose input channels are not a
    multiple of 8, SPADE layers below 16 normalised / 8 map channels) and the nn.Linear weight generators (generator.py:103-110),
    i.e. places where the definition is MORE precise; (4) every operation on the policy's fp32 side that occurs on this path
    (spectral-norm power iteration, loss reductions, grid_sample, pooling) is fp32 in the definition too - it computes nothing
    but the routed convolutions in half; (5) element-wise / normalisation operations that autocast runs in half only because
    their INPUT arrived as half (batch_norm, leaky_relu, add, mul, upsample, tanh ...) are fp32 in the definition (it keeps
    convolution outputs fp32, like the product's kernels: fp32 accumulator and epilogue) - listed, not asserted equal."""
    from collections import Counter
    from oracle import np_oracle as NO
    from oracle.op_census import census_mode
    with open(os.path.join(GOLD, 'autocast_ops.json')) as f:
        gold = json.load(f)[name]
    opt = _opt_from_flags(gold['flags'])
    model = mc._model().create_model(opt)
    sdG, sdD = mc.fill_state(model.netG), mc.fill_state(model.netD)
    nl = opt.label_nc if opt.label_nc != 0 else opt.input_nc
    h, w = int(opt.fineSize / opt.aspect_ratio), opt.fineSize
    b = opt.batchSize
    data = mc.synth_street_inputs(b, h, w, 4242, opt.label_nc) if opt.label_nc != 0 else mc.synth_pose_inputs(b, h, w, 4242, nl)
    cfg = O.cfg_from_opt(opt)
    routed = []            # (weight shape, ran on half operands) of every convolution the installed arithmetic was asked for

    def recording(x, wt, bias, stride, padding, dx_half=False, per_sample=False):
        cout, cin = wt.shape[0], wt.shape[1]
        routed.append((tuple(wt.shape), int(stride), not (cout % 8 != 0 or (per_sample and cin % 8 != 0))))
        return NO.amp_conv2d(x, wt, bias, stride, padding, dx_half, per_sample)
    cen = census_mode()
    with torch.no_grad(), O.arithmetic(recording), cen:
        O.g_step_losses(sdG, sdD, cfg, *data)
    # (1) the same layers
    key = lambda c: (tuple(c[0]), int(c[1]))
    ref_convs = Counter(key(c) for c in gold['convs'])
    ours_all = Counter(key(c) for c in cen.convs)
    assert ours_all == ref_convs, (ours_all - ref_convs, ref_convs - ours_all)
    ref_mm = Counter(repr(m[1]) for m in gold['mms'])
    ours_mm = Counter(repr(m[1]) for m in cen.mms)
    assert ours_mm == ref_mm, (ours_mm - ref_mm, ref_mm - ours_mm)
    # (2) half in the definition -> half under autocast
    auto_half = Counter(key(c) for c in gold['convs'] if c[2] == 'float16')
    ours_half = Counter((shape, st) for shape, st, half in routed if half)
    assert not (ours_half - auto_half), ours_half - auto_half
    # (3) the complement is the documented set
    kept_fp32 = auto_half - ours_half
    for (shape, st), n in kept_fp32.items():
        cout, cin = shape[0], shape[1]
        assert cout % 8 != 0 or cin % 8 != 0 or shape[2:] == (1, 1), ('the definition keeps %s fp32 without a stated reason' % (shape,))
    assert all(m[2] == 'float16' for m in gold['mms'])                  # autocast: every nn.Linear in half ...
    assert all(d == {'float32': n} for k, v in cen.ops.items() if k in ('addmm', 'mm', 'bmm') for d, n in [(v, sum(v.values()))])   # ... the definition: fp32
    # (4) the policy's fp32 side
    for op, dt in gold['ops'].items():
        if set(dt) == {'float32'} and op in cen.ops:
            assert set(cen.ops[op]) == {'float32'}, (op, cen.ops[op])
    for op in ('grid_sampler_2d', 'mv', 'dot', 'linalg_vector_norm', 'abs', 'avg_pool2d'):
        if op in gold['ops']:
            assert set(gold['ops'][op]) == {'float32'}, (op, gold['ops'][op])
    # (5) everything that is not a routed convolution is fp32 in the definition
    assert all(set(v) <= {'float32'} for k, v in cen.ops.items() if k != 'convolution'), {k: v for k, v in cen.ops.items() if set(v) - {'float32'}}


@pytest.mark.parametrize('case', ['pose_fullsize', 'street_fullsize', 'pose_face_d_fullsize'])
def test_fullsize_fixture_is_consistent(case):
    """The full-size fixtures (BASELINE configs[2] / [4] at full width and resolution; oracle/make_golden.py `fullsize`) hold the
    unmodified reference's fp32 iteration AND its distance to the oracle's fp64 evaluation of the same iteration, computed when the
    fixture was minted (minutes of host time - not repeated here): the two agree to rounding, i.e. the oracle reproduces the
    reference at the benchmarked size too, and every stored record is complete."""
    g = _load(case)
    opt = _opt_from_flags(g['flags'].replace('--loadSize', '--loadSize'))
    M = mc._model()
    with torch.device('meta'):
        model = M.create_model(opt)
    out = g['outputs']
    assert out['fake']['noise_l2'] <= 1e-4 * out['fake']['norm'], out['fake']
    assert out['fake']['sketch'].shape == (mc.SKETCH_K_IMAGE,)
    for i, ref64 in enumerate(g['d_losses64']):
        assert abs(g['d_losses'][i] - ref64) <= 1e-5 * max(1.0, abs(ref64))
    for k, ref64 in g['g_losses64'].items():
        ref = g['g_losses'][g['loss_names'].index(k)]
        assert abs(ref - ref64) <= 1e-5 * max(1.0, abs(ref64)), (k, ref, ref64)
    for net, rec in ((model.netG, g['grad_G']), (model.netD, g['grad_D'])):
        names = {n for n, p in net.named_parameters()}
        assert set(rec) <= names and len(rec) >= 0.9 * len(names), (len(rec), len(names))
        norms = sorted(v['norm'] for v in rec.values())
        floor = 1e-2 * norms[len(norms) // 2]
        for k, v in rec.items():
            assert v['sketch'].shape == (mc.SKETCH_K_GRAD,)
            # the sketch of a tensor is consistent with its norm (E ||sketch||^2 = ||x||^2; 16 buckets)
            assert float(v['sketch'].norm()) <= 3.0 * v['norm'] + 1e-30, k
            # fp32 reference vs fp64 oracle: rounding noise (sums of 10^5 ... 10^6 terms with cancellation) and the few activations a
            # 3e-6 image difference moves across a LeakyReLU / hinge kink: up to 7e-3 of the norm in these fixtures
            assert v['noise_l2'] <= 2e-2 * max(v['norm'], floor), (k, v['noise_l2'], v['norm'])
