"""Eager G forward+backward timing at BASELINE config C3 (pose 512x512, B=2, adaptive_spade+warp_ref+spade_combine)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import model_checks as mc
net = mc._net()
dev = torch.device('cuda:0')
B = int(os.environ.get('B', 2)); S = int(os.environ.get('S', 512))
opt = mc.make_opt(fineSize=S, loadSize=S, warp_ref=True, spade_combine=True)
torch.manual_seed(0)
G = net.define_G(opt).to(dev).train()
print('params', sum(p.numel() for p in G.parameters()))
tl, ti, rl, ri = mc.synth_pose_inputs(B, S, S, 1)
label, rl, ri = tl[:, 0].to(dev), rl.to(dev), ri.to(dev)
def step():
    out = G(label, rl, ri, [None, None])
    loss = out[0].mean() + out[1][0].mean() * 1e-3 + out[2][0].mean()
    loss.backward()
    return out
for _ in range(2): step()
torch.cuda.synchronize()
t = time.time(); n = 3
for _ in range(n):
    with torch.no_grad(): G(label, rl, ri, [None, None])
torch.cuda.synchronize(); tf = (time.time() - t) / n
t = time.time()
for _ in range(n): step()
torch.cuda.synchronize(); tfb = (time.time() - t) / n
print(json.dumps(dict(B=B, S=S, fwd_ms=round(tf * 1e3, 2), fwd_bwd_ms=round(tfb * 1e3, 2), mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 2),
                      fwd_tflops=round(386.6e9 * B / tf / 1e12, 2))))
