"""Library launches of one steady-state training iteration by kernel, counted by the SIMT emulator on the CPU (C3 flags, tiny
widths: the count depends on the structure of the step, not on tensor sizes).  python tools/launch_report.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['FSV2V_EMU'] = '1'
import torch  # noqa: E402
import model_checks as mc  # noqa: E402
from importlib import import_module  # noqa: E402

M = mc._model()
build = import_module('few-shot-vid2vid_amd.build')
build.build_emu()
lib = import_module('few-shot-vid2vid_amd.lib')
opt = mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, ngf=int(os.environ.get('NGF', '16')), nff=16, ndf=8)
model = M.create_model(opt)
mc.fill_state(model.netG); mc.fill_state(model.netD)
model.train()
opt_G, opt_D = model.build_optimizers()
tl, ti, rl, ri = mc.synth_pose_inputs(2, 64, 64, 900, opt.input_nc)
data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
h = lib.get_lib()
h.fsv_emu_launch_report.restype = ctypes.c_int
buf = ctypes.create_string_buffer(1 << 16)
for it in range(3):
    h.fsv_emu_launch_report(buf, len(buf), 1)
    M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
    nD = sum(int(l.rsplit(' ', 1)[1]) for l in (h.fsv_emu_launch_report(buf, len(buf), 0), buf.value.decode())[1].splitlines())
    g, _, _ = model(data, mode='generator')
    M.loss_backward(opt, g, opt_G, 0)
h.fsv_emu_launch_report(buf, len(buf), 0)
rows = sorted(((int(l.rsplit(' ', 1)[1]), l.rsplit(' ', 1)[0]) for l in buf.value.decode().splitlines()), reverse=True)
print('iteration 2: %d library launches (D step %d)' % (sum(r[0] for r in rows), nD))
for n, name in rows:
    print('%5d  %s' % (n, name))
