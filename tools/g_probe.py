"""SPADE-generator forward at 512x512 (north-star target: >= 40 % MFMA utilisation at bs = 8 on one MI355X).

Times the generator forward (training-mode statistics, no autograd tape) of BASELINE config C3 flags
(adaptive_spade + warp_ref + spade_combine: 386.6 GFLOP / frame) and of adaptive_spade only (179.2 GFLOP / frame),
eagerly and as a replayed hipGraph, and prints algorithmic TFLOP/s against the 157.3 TFLOP/s fp32 matrix peak."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import model_checks as mc
net = mc._net()
dev = torch.device('cuda:0')
S = int(os.environ.get('S', 512))
PEAK = 157.3


def run(B, combine):
    opt = mc.make_opt(fineSize=S, loadSize=S, warp_ref=combine, spade_combine=combine)
    torch.manual_seed(0)
    G = net.define_G(opt).to(dev).train()
    tl, ti, rl, ri = mc.synth_pose_inputs(B, S, S, 1)
    label, rl, ri = tl[:, 0].to(dev), rl.to(dev), ri.to(dev)
    gflop = (386.6 if combine else 179.2) * B * (S / 512.0) ** 2

    def fwd():
        with torch.no_grad():
            return G(label, rl, ri, [None, None])
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fwd()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    t = time.time(); n = 5
    for _ in range(n): fwd()
    torch.cuda.synchronize(); te = (time.time() - t) / n
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fwd()
    g.replay(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); tg = (time.time() - t) / n
    print(json.dumps(dict(case='G fwd %s' % ('warp+combine' if combine else 'adaptive_spade only'), B=B, S=S,
                          eager_ms=round(te * 1e3, 2), graph_ms=round(tg * 1e3, 2), gflop=round(gflop, 1),
                          tflops=round(gflop / tg / 1e3, 2), frac_fp32_mfma_peak=round(gflop / tg / 1e3 / PEAK, 4),
                          mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))), flush=True)
    del g, G
    torch.cuda.empty_cache()


for B in (2, 8):
    for combine in (True, False):
        run(B, combine)
