import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import model_checks as mc
from oracle import fsv_oracle as O
dev = torch.device('cuda:0') if torch.cuda.is_available() else torch.device('cpu')
opt = mc.tiny_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=128, loadSize=128)
M = mc._model()
model = M.create_model(opt)
sdG0, sdD0 = mc.fill_state(model.netG), mc.fill_state(model.netD)
model = model.to(dev).train()
opt_G, opt_D = model.build_optimizers(); opt_G.set_lr(0.0); opt_D.set_lr(0.0)
data = mc.synth_pose_inputs(2, 128, 128, 21)
cfg = O.cfg_from_opt(opt)
# oracle fake image (fp32) from the D step's no-grad G forward
sdG = {k: v.clone() for k, v in sdG0.items()}
with torch.no_grad():
    g = O.step_generate(sdG, cfg, *data)
tl, ti, rl, ri = [t.to(dev) for t in data]
with torch.no_grad():
    (fake, raw, _, _, _), _, _, _ = model.generate_images(tl, ti, rl, ri, [None, None, None])
d = (fake.cpu() - g['fake']).abs()
print('fake diff max %.3e mean %.3e ref max %.3e' % (d.max(), d.mean(), g['fake'].abs().max()))
# D alone on identical inputs: feed the ORACLE's fake to both
lab4 = data[0].reshape(-1, *data[0].shape[-3:])
real = data[1][:, 0]
sdD = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(('_u','_v'))) for k, v in sdD0.items()}
lo = O._discriminate(sdD, cfg, lab4, g['fake'], real, g['ref_label'], g['ref_image'], True)
(lo[0] + lo[1]).backward()
lc = model.lossCollector
losses = lc.gan_losses(model.netD, tl, [real.to(dev), None], [g['fake'].to(dev), None], g['ref_label'].to(dev), g['ref_image'].to(dev), True)
for p in model.netD.parameters(): p.grad = None
(losses[0] + losses[1]).sum().backward()
print('losses', [float(x) for x in lo], [float(x) for x in losses[:2]])
rows = []
for n, p in model.netD.named_parameters():
    r = sdD[n].grad
    rows.append((float((p.grad.cpu() - r).abs().max() / max(r.abs().max(), 1e-12)), n, float(r.abs().max())))
rows.sort(reverse=True)
for r in rows[:6]: print('%.2e %s scale %.2e' % r)
