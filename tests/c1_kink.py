"""Where the two outcomes of the C1 full-size step come from (tests/c1_repro.py: the gradient of ref_img_up_2.conv.weight is either
~2e-3 or exactly 3.58e-2 away from the oracle): the pre-activation values z = BatchNorm(conv(x)) of that layer - and of every other
conv -> BatchNorm -> LeakyReLU layer of the reference-image encoder - are listed by how close they sit to the LeakyReLU kink,
for one run with FSV_DETERMINISTIC=1 (always the 3.58e-2 outcome) and several default runs.

    python tests/c1_kink.py          (on the GPU box)
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import model_checks as mc

dev = torch.device('cuda:0')
M = mc._model()
opt = mc.make_opt(dataset_mode='fewshot_face', input_nc=1, fineSize=128, loadSize=128, batchSize=1)
b, seed = 1, 21
h, w = int(opt.fineSize / opt.aspect_ratio), opt.fineSize
nl = opt.input_nc
data = mc.synth_pose_inputs(b, h, w, seed, nl)
data = mc.with_n_shot(data, opt.n_shot, b, h, w, seed, nl)


def run(det):
    os.environ['FSV_DETERMINISTIC'] = '1' if det else '0'
    torch.manual_seed(0)
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    model = model.to(dev).train()
    opt_G, opt_D = model.build_optimizers()
    opt_G.set_lr(0.0); opt_D.set_lr(0.0)
    caps = {}

    def hook(name):
        def f(mod, inp, out):
            if name not in caps:          # the generator step is the second forward of the iteration: keep the LAST call
                pass
            caps[name] = out.detach().double().cpu().clone()
        return f
    names = [n for n, m in model.netG.named_modules() if n.startswith('ref_img_') and n.endswith('.conv')]
    for n in names:
        dict(model.netG.named_modules())[n].register_forward_hook(hook(n))
    tl, ti, rl, ri = [t.to(dev) for t in data]
    data_list = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
    M.loss_backward(opt, model(data_list, mode='discriminator'), opt_D, 1)
    g_losses, generated, prev = model(data_list, save_images=True, mode='generator')
    M.loss_backward(opt, g_losses, opt_G, 0)
    zs = {}
    mods = dict(model.netG.named_modules())
    for n, y in caps.items():
        bn = mods[n[:-5] + '.bn']
        yy = y.permute(1, 0, 2, 3).reshape(y.shape[1], -1)                 # [C, N*H*W]
        mu, var = yy.mean(1, keepdim=True), yy.var(1, unbiased=False, keepdim=True)
        z = (yy - mu) / torch.sqrt(var + 1e-5) * bn.weight.detach().double().cpu()[:, None] + bn.bias.detach().double().cpu()[:, None]
        zs[n] = z
    g = model.netG.ref_img_up_2.conv.weight_orig.grad.detach().double().cpu()
    return zs, g


zd, gd = run(True)
print(json.dumps({'layer': 'shapes', **{n: list(z.shape) for n, z in zd.items()}}))
for n, z in zd.items():
    a = z.abs()
    print(json.dumps({'det run': n, 'pixels': z.shape[1], 'min|z|': float(a.min()), 'n(|z|<1e-4)': int((a < 1e-4).sum()),
                      'n(|z|<1e-3)': int((a < 1e-3).sum()), 'n': a.numel()}))
for k in range(6):
    z0, g0 = run(False)
    rel = float((g0 - gd).norm() / gd.norm())
    flips = {}
    for n in zd:
        diff = (torch.sign(z0[n]) != torch.sign(zd[n]))
        if int(diff.sum()):
            idx = diff.nonzero()
            flips[n] = [(int(c), int(p), float(zd[n][c, p]), float(z0[n][c, p])) for c, p in idx[:6].tolist()]
    print(json.dumps({'default run': k, '|g - g_det| / |g_det| of ref_img_up_2': round(rel, 5), 'sign flips (channel, pixel, z det, z default)': flips}), flush=True)
