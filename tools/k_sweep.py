"""Short-K 1x1 layers at full resolution (M = 524288 pixels, Cout 32 / 64): isolated, warm timing of the gather-GEMM launch the plan
picks, K swept - what the in-step 100 - 160 us of the K64 / K96 layers are made of (tools/shape_profile.py lists them)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fsv2v_amd  # noqa
from importlib import import_module
conv = import_module('few-shot-vid2vid_amd.conv')
dev = torch.device('cuda:0')
NREP = 20
for cout in (32, 64):
    for cin in (32, 64, 96, 128, 160, 256):
        n, h, w = 2, 512, 512
        x = conv.to_nhwc(torch.randn(n, cin, h, w, device=dev)); wt = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
        g = conv.Geom(1, 1, 1, 0)
        wf, kpad, ldw = conv.prep_weight(wt, 0, g)
        f = lambda: conv.conv_forward(x, wf, ldw, cout, g)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f(); f()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(NREP):
                f()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / NREP)
        us = sorted(ts)[2]
        mb = n * h * w * (cin + cout) * 4 / 1e6
        print(json.dumps(dict(M=n * h * w, N=cout, K=cin, us=round(us, 1), MB=round(mb, 1), TBps=round(mb / us, 2))), flush=True)
