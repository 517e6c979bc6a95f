"""A/B of the half-precision convolution kernels (csrc/conv_h.hip) on layer shapes of the C5 / C3 steps: forward tiles x LDS buffer
counts x K splits and the weight-gradient tiles, each captured as a hipGraph of 20 back-to-back launches, replayed interleaved,
median of 5.  Output: algorithmic TFLOP/s per configuration ('auto' = fsv_hconv_plan), the fp32 kernel's plan beside it."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fsv2v_amd  # noqa
from importlib import import_module
conv = import_module('few-shot-vid2vid_amd.conv')
hc = import_module('few-shot-vid2vid_amd.hconv')
dev = torch.device('cuda:0')
shapes = [('M8192 N256 K2304', 1, 256, 64, 128, 256, 3), ('M32768 N128 K1152', 1, 128, 128, 256, 128, 3),
          ('M131072 N64 K576', 1, 64, 256, 512, 64, 3), ('M524288 N32 K288', 1, 32, 512, 1024, 32, 3),
          ('M2048 N512 K4608', 1, 512, 32, 64, 512, 3), ('M512 N1024 K9216', 1, 1024, 16, 32, 1024, 3),
          ('M8192 N256 K4608', 1, 512, 64, 128, 256, 3), ('M32768 N128 K2304', 1, 256, 128, 256, 128, 3),
          ('M131072 N64 K1152', 1, 128, 256, 512, 64, 3), ('M524288 N32 K576', 1, 64, 512, 1024, 32, 3),
          ('M32768 N256 K128', 1, 128, 128, 256, 256, 1), ('M1024 N1024 K1024', 1, 1024, 1, 1024, 1024, 1),
          ('M131072 N64 K1024', 2, 64, 257, 257, 64, 4), ('M131072 N64 K288', 1, 32, 256, 512, 64, 3),
          ('M32768 N128 K576', 1, 64, 128, 256, 128, 3), ('M131072 N128 K64', 1, 64, 256, 512, 128, 1)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[0] for a in sys.argv[1:])]
NREP = 20
WHAT = os.environ.get('FSV_HAB', 'fwd,wgrad').split(',')


def graph_of(f):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f(); f()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(NREP):
            f()
    return gr


def measure(graphs, flops):
    res = {c: [] for c in graphs}
    for rnd in range(5):
        for c, gr in graphs.items():
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[c].append(flops / (e0.elapsed_time(e1) / NREP * 1e-3) / 1e12)
    return {c: round(sorted(v)[len(v) // 2], 1) for c, v in res.items()}


for name, n, cin, h, w, cout, k in shapes:
    x = conv.to_nhwc(torch.randn(n, cin, h, w, device=dev)); wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    g = conv.Geom(k, k, 1, k // 2)
    wf, kpad, ldw = conv.prep_weight(wt, 0, g)
    wh, kp, nrows = hc.prep_weight_h(wf)
    xh = hc.to_half_nhwc(x)
    flops = 2.0 * n * h * w * cout * cin * k * k
    oh, ow = g.out_hw(h, w)
    if 'fwd' in WHAT:
        graphs = {'f32 auto': graph_of(lambda: conv.conv_forward(x, wf, ldw, cout, g, bias=b, act=conv.ACT_LRELU))}
        for half in (True, False):
            graphs['h auto %s' % ('h' if half else 'f')] = graph_of(lambda: hc.conv_forward_h(xh, wh, kp, nrows, cout, g, bias=b, act=conv.ACT_LRELU, out_half=half))
        if 'epi' in WHAT:
            r = conv.to_nhwc(torch.randn(n, cout, oh, ow, device=dev))
            graphs['h f +res'] = graph_of(lambda: hc.conv_forward_h(xh, wh, kp, nrows, cout, g, bias=b, res=r, out_half=False))

            def with_stats():
                with conv.stats_pass(dev):
                    for _ in range(NREP):
                        hc.conv_forward_h(xh, wh, kp, nrows, cout, g, bias=b, out_half=False, stats={'groups': 1})
            for _ in range(2):
                with_stats()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                with conv.stats_pass(dev):
                    for _ in range(NREP):
                        hc.conv_forward_h(xh, wh, kp, nrows, cout, g, bias=b, out_half=False, stats={'groups': 1})
            graphs['h f +stats'] = gr
        for t in ((0, 1, 2, 3, 4, 5, 9, 16, 17, 18, 19, 20, 21, 25) if 'tiles' in WHAT or 'epi' not in WHAT else ()):
            bn = {0: 128, 1: 64, 2: 32, 3: 128, 4: 64, 5: 128, 9: 128}[t & 15]
            if (bn == 128 and cout < 128) or (bn == 64 and cout < 64) or (bn == 32 and cout > 32):
                continue
            for sp in ((1, 2, 4) if 'splits' in WHAT else (1,)):
                if sp > 1 and (n * oh * ow // 64) * (-(-cout // bn)) > 600:
                    continue
                graphs['t%d/s%d' % (t, sp)] = graph_of(lambda: hc.conv_forward_h(xh, wh, kp, nrows, cout, g, bias=b, act=conv.ACT_LRELU, out_half=True, force_tile=t, force_split=sp))
        print(json.dumps({'case': name, 'kind': 'fwd', **measure(graphs, flops)}), flush=True)
    if 'wgrad' in WHAT and hc.wgrad_eligible(cin, cout, oh, ow):
        dy = conv.to_nhwc(torch.randn(n, cout, oh, ow, device=dev)); dyh = hc.to_half_nhwc(dy)
        graphs = {'f32 auto': graph_of(lambda: conv.conv_wgrad(x, dy, g, (cout, cin, k, k), raw=True))}
        for t in (0, 1, 2, 3, 4, 5, 6):
            bn = {0: 0, 1: 64, 2: 64, 3: 128, 4: 128, 5: 32, 6: 64}[t]
            if (bn == 128 and cout < 128) or (bn == 64 and cout < 64) or (bn == 32 and cout > 32):
                continue
            for sp in ((0,) if t == 0 else (0, 4, 16)):
                graphs['t%d/s%d' % (t, sp)] = graph_of(lambda: hc.conv_wgrad_h(xh, dyh, g, force_tile=t, force_split=sp))
        print(json.dumps({'case': name, 'kind': 'wgrad', **measure(graphs, flops)}), flush=True)
