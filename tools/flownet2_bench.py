"""FlowNet2 teacher throughput on one MI355X: full-width network (162.5 M random parameters), image pairs at 512x512."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fsv2v_amd  # noqa
from importlib import import_module
fn = import_module('few-shot-vid2vid_amd.flownet2')
dev = torch.device('cuda:0')
b, size = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 512
net = fn.FlowNet2().to(dev).eval()
x = torch.rand(b, 3, 2, size, size, device=dev)
for _ in range(2):
    net(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    net(x)
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g):
        y = net(x)
    run = g.replay
    mode = 'hipgraph'
except Exception as e:            # noqa
    run = lambda: net(x)
    mode = 'eager (%s)' % type(e).__name__
run(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
print({'pairs_per_s': round(b / dt, 2), 'ms_per_batch': round(dt * 1e3, 2), 'batch': b, 'size': size, 'mode': mode})
