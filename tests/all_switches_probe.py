"""One tiny training iteration under whatever FSV_* switches the environment sets; prints the losses and an image checksum as one
JSON line (driver: test_all_switches_emu.py; on a GPU: python tests/all_switches_probe.py cuda)."""
import json
import sys

import torch

import model_checks as mc

dev = torch.device(sys.argv[1] if len(sys.argv) > 1 else 'cpu')
M = mc._model()
opt = mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, ngf=16, nff=16, fineSize=32, loadSize=32, n_downsample_G=3,
                  n_adaptive_layers=2)
model = M.create_model(opt)
mc.fill_state(model.netG); mc.fill_state(model.netD)
model = model.to(dev).train()
opt_G, opt_D = model.build_optimizers()
tl, ti, rl, ri = [t.to(dev) for t in mc.synth_pose_inputs(2, 32, 32, 904, opt.input_nc)]
data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
for it in range(2):
    d = M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
    g, gen, _ = model(data, save_images=True, mode='generator')
    g = M.loss_backward(opt, g, opt_G, 0)
print(json.dumps(dict(d=[float(x.detach()) for x in d], g=[float(x.detach()) for x in g if not isinstance(x, int)],
                      img=gen[0].detach().double().cpu().flatten()[::37].tolist())))
