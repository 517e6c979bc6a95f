"""test.py-style inference throughput on one MI355X: eval-mode generator (C3 flags, 512x512), one frame per call, previous
frame fed back (temporal branch initialised), frame-0 generated weights re-used (opt.isTrain False)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from importlib import import_module
import fsv2v_amd  # noqa
M = import_module('few-shot-vid2vid_amd.model')
dev = torch.device('cuda:0')
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt = bench.build_opt(512, b)
model = M.create_model(opt).to(dev)
model.init_temporal_model()
model = model.to(dev)
# a few training-mode passes so that the eval-mode statistics / spectral vectors are not the initial ones
data = bench.make_data(b, 512, 7, dev)
with torch.no_grad():
    for _ in range(3):
        model(data, mode='generator')
model.eval()
opt.isTrain = False
tl, ref_l, ref_i = data[0], data[4], data[5]
frame = [tl, None, None, None, ref_l, ref_i, None, None, None]
for _ in range(3):
    out = model(frame)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = model(frame)            # steady state: t > 0, prevs are persistent tensors of the same shapes
    run, mode = g.replay, 'hipgraph (steady-state frame)'
except Exception as e:                # noqa
    run, mode = (lambda: model(frame)), 'eager (%s)' % type(e).__name__
run(); torch.cuda.synchronize()
t = time.perf_counter()
n = 30
for _ in range(n):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print({'frames_per_s': round(b / dt, 2), 'ms_per_frame_batch': round(dt * 1e3, 2), 'batch': b, 'mode': mode})
