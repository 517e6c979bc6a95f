"""A/B of the operand modes (exact fp32 / f16 / bf16x3, csrc/conv_np.hip) on the layer shapes of the bench step at B=2:
forward gather-GEMM and weight gradient, each captured as a hipGraph of 20 back-to-back launches, replayed interleaved,
median of 5.  Output: algorithmic TFLOP/s per mode (auto plan) and the max relative error of the narrow results against the
fp32 kernel.  First thing to run when the narrow kernels meet the hardware:

    python tools/np_ab.py > gpurun_out/np_ab.jsonl
"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fsv2v_amd  # noqa
from importlib import import_module
conv = import_module('few-shot-vid2vid_amd.conv')
dev = torch.device('cuda:0')
MODES = [(0, 'f32'), (1, 'f16'), (2, 'bf16x3')]
shapes = [('M8192 N256 K2304', 2, 256, 64, 64, 256, 3), ('M32768 N128 K576', 2, 64, 128, 128, 128, 3),
          ('M32768 N128 K1152', 2, 128, 128, 128, 128, 3), ('M2048 N512 K9216', 2, 1024, 32, 32, 512, 3),
          ('M131072 N64 K288', 2, 32, 256, 256, 64, 3), ('M1024 N512 K512', 1, 512, 1, 1024, 512, 1),
          ('M512 N1024 K4608', 2, 512, 16, 16, 1024, 3), ('M8192 N128 K512', 2, 512, 64, 64, 128, 1),
          ('M32768 N256 K1152', 2, 128, 128, 128, 256, 3), ('M131072 N128 K576', 2, 64, 256, 256, 128, 3),
          ('M8192 N256 K9216', 2, 1024, 64, 64, 256, 3), ('M8192 N1024 K2304', 2, 256, 64, 64, 1024, 3)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[0] for a in sys.argv[1:])]
NREP = 20


def capture(f):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = f(); f()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(NREP):
            f()
    return gr, out


for name, n, cin, h, w, cout, k in shapes:
    x = conv.to_nhwc(torch.randn(n, cin, h, w, device=dev)); wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    g = conv.Geom(k, k, 1, k // 2)
    wf, kpad, ldw = conv.prep_weight(wt, 0, g)
    dy = conv.to_nhwc(torch.randn(n, cout, h, w, device=dev))
    flops = 2.0 * n * h * w * cout * cin * k * k
    graphs, outs = {}, {}
    for mode, mname in MODES:
        conv.set_mfma_mode(mode)
        graphs[('fwd', mname)], outs[('fwd', mname)] = capture(
            lambda: conv.conv_forward(x, wf, ldw, cout, g, bias=b, act=conv.ACT_LRELU))
        graphs[('wgrad', mname)], outs[('wgrad', mname)] = capture(
            lambda: conv.conv_wgrad(x, dy, g, (cout, cin, k, k), raw=True))
    conv.set_mfma_mode(0)
    res = {c: [] for c in graphs}
    for rnd in range(5):
        for c, gr in graphs.items():
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[c].append(flops / (e0.elapsed_time(e1) / NREP * 1e-3) / 1e12)
    row = {'case': name}
    for (op, mname), v in res.items():
        row['%s_%s' % (op, mname)] = round(sorted(v)[len(v) // 2], 1)
        if mname != 'f32':
            # the K-major weight-gradient buffer is padded to multiples of 32 and only [:K, :Cout] is ever written
            crop = (lambda t: t[..., :k * k * cin, :cout]) if op == 'wgrad' else (lambda t: t)
            ref, got = crop(outs[(op, 'f32')]), crop(outs[(op, mname)])
            row['%s_%s_err' % (op, mname)] = float('%.2e' % float((got - ref).abs().max() / ref.abs().max()))
    print(json.dumps(row), flush=True)
