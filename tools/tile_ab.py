"""A/B of conv tile / split-K configurations on layer shapes of the bench step at B=2.

Every configuration is captured as a hipGraph of 20 back-to-back launches (memset + GEMM + bias/activation pass when
split-K is on, exactly what the product path issues) and the configurations are replayed interleaved, median of 5.
Output: algorithmic TFLOP/s per configuration; 'auto' is what fsv_conv_plan picks today."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fsv2v_amd  # noqa
from importlib import import_module
conv = import_module('few-shot-vid2vid_amd.conv')
dev = torch.device('cuda:0')
# FSV_AB_EXPERIMENTAL=1 adds the force_tile-only variants (10 / 11 / 12: 64x128 / 128x128 / 128x64 with a prefetch distance of two
# chunks) next to the tiles of the plan
EXPERIMENTAL = (10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 21, 22, 27) if os.environ.get('FSV_AB_EXPERIMENTAL', '0') == '1' else ()      # 13 - 15: mid-chunk barrier
ONLY = tuple(int(t) for t in os.environ.get('FSV_AB_TILES', '').split(',') if t)          # restrict the tile ids (short runs)
SPLITS = tuple(int(t) for t in os.environ.get('FSV_AB_SPLITS', '1,2,4,8').split(','))
cfgs = [(-1, 0)] + [(t_, s_) for t_ in (0, 1, 2, 4, 9) + EXPERIMENTAL if not ONLY or t_ in ONLY for s_ in SPLITS]
shapes = [('M8192 N256 K2304', 2, 256, 64, 64, 256, 3), ('M32768 N128 K576', 2, 64, 128, 128, 128, 3),
          ('M32768 N128 K1152', 2, 128, 128, 128, 128, 3), ('M32768 N128 K2304', 2, 256, 128, 128, 128, 3),
          ('M2048 N512 K9216', 2, 1024, 32, 32, 512, 3), ('M2048 N512 K2304', 2, 256, 32, 32, 512, 3),
          ('M8192 N256 K1152', 2, 128, 64, 64, 256, 3), ('M131072 N64 K288', 2, 32, 256, 256, 64, 3),
          ('M1024 N512 K512', 1, 512, 1, 1024, 512, 1), ('M512 N1024 K4608', 2, 512, 16, 16, 1024, 3),
          ('M8192 N128 K512', 2, 512, 64, 64, 128, 1), ('M32768 N64 K256', 2, 256, 128, 128, 64, 1),
          ('M32768 N256 K1152', 2, 128, 128, 128, 256, 3), ('M131072 N128 K576', 2, 64, 256, 256, 128, 3),
          ('M8192 N256 K4608', 2, 512, 64, 64, 256, 3), ('M8192 N256 K9216', 2, 1024, 64, 64, 256, 3),
          ('M2048 N1024 K4608', 2, 512, 32, 32, 1024, 3), ('M8192 N1024 K2304', 2, 256, 64, 64, 1024, 3),
          ('M32768 N128 K4608', 2, 512, 128, 128, 128, 3), ('M512 N1024 K9216', 2, 1024, 16, 16, 1024, 3),
          # awkward workgroup counts (discriminator 33 x 66 / 17 x 34 maps, k4 s1 p2)
          ('M4356 N256 K8192', 2, 512, 32, 65, 256, 4), ('M4624 N512 K4096', 2, 256, 33, 67, 512, 4),
          ('M2048 N512 K9216', 2, 1024, 32, 32, 512, 3),
          # full-resolution thin layers (Cout 32)
          ('M524288 N32 K576', 2, 64, 512, 512, 32, 3), ('M524288 N32 K288', 2, 32, 512, 512, 32, 3),
          ('M131072 N32 K128', 2, 128, 256, 256, 32, 1)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[0] for a in sys.argv[1:])]
NREP = 20
for name, n, cin, h, w, cout, k in shapes:
    x = conv.to_nhwc(torch.randn(n, cin, h, w, device=dev)); wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    g = conv.Geom(k, k, 1, k // 2)
    wf, kpad, ldw = conv.prep_weight(wt, 0, g)
    flops = 2.0 * n * h * w * cout * cin * k * k
    graphs = {}
    for c in cfgs:
        if c[0] in (0, 9, 10, 11, 13, 14, 16, 21) and cout < 128 or c[0] in (1, 12, 15, 22) and cout < 64 or c[0] in (2, 18) and cout > 32:
            continue
        f = lambda: conv.conv_forward(x, wf, ldw, cout, g, bias=b, act=conv.ACT_LRELU, force_tile=c[0], force_split=c[1])
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f(); f()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(NREP):
                f()
        graphs[c] = gr
    if os.environ.get('FSV_AB_STATS', '0') == '1' and cin % 4 == 0:
        # the plan's configuration with the BatchNorm statistics of the output taken in the epilogue (fp64 atomics into 32 slots)
        f = lambda: conv.conv_forward(x, wf, ldw, cout, g, bias=b, act=conv.ACT_NONE, stats={'groups': 1})
        for _ in range(2):
            with conv.stats_pass(dev):
                for _ in range(NREP):
                    f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            with conv.stats_pass(dev):
                for _ in range(NREP):
                    f()
        graphs[('stats', 0)] = gr
    res = {c: [] for c in graphs}
    for rnd in range(5):
        for c, gr in graphs.items():
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[c].append(flops / (e0.elapsed_time(e1) / NREP * 1e-3) / 1e12)
    print(json.dumps({'case': name, **{('auto' if c[0] == -1 else 'auto+stats' if c[0] == 'stats' else 't%d/s%d' % c): round(sorted(v)[len(v) // 2], 1)
                                       for c, v in res.items()}}), flush=True)
