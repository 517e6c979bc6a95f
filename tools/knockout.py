"""What each part of the dominant gather-GEMM loop costs: the 64x128 / 8-wave / prefetch-2 / in-place-fragment kernel (tile id 13)
beside forms of itself with one part left out (library built with -DFSV_DIAG: few-shot-vid2vid_amd.build.build_hip_diag(), loaded
through FSV2V_LIB).  The knock-out kernels compute garbage; only their time is read.  Same harness as tools/tile_ab.py:
20 launches per hipGraph, configurations replayed interleaved, median of 5.

    python -c "import fsv2v_amd; from importlib import import_module as im; im('few-shot-vid2vid_amd.build').build_hip_diag()"
    FSV2V_LIB=tools/_diag/libfsv2v_hip_diag.so python tools/knockout.py            (on the GPU box)
"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fsv2v_amd  # noqa
from importlib import import_module
conv = import_module('few-shot-vid2vid_amd.conv')
dev = torch.device('cuda:0')
NAMES = {13: 'full', 30: 'no barrier', 31: 'no LDS stores', 32: 'no global loads', 33: 'one fragment read per chunk',
         34: 'no offset arithmetic', 35: 'MFMA + barrier only', 36: 'MFMA only', 37: 'no epilogue stores'}
shapes = [('M8192 N256 K2304', 2, 256, 64, 64, 256, 3), ('M32768 N128 K2304', 2, 256, 128, 128, 128, 3),
          ('M8192 N256 K9216', 2, 1024, 64, 64, 256, 3)]
NREP = 20
for name, n, cin, h, w, cout, k in shapes:
    x = conv.to_nhwc(torch.randn(n, cin, h, w, device=dev)); wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    g = conv.Geom(k, k, 1, k // 2)
    wf, kpad, ldw = conv.prep_weight(wt, 0, g)
    flops = 2.0 * n * h * w * cout * cin * k * k
    graphs = {}
    for t in NAMES:
        f = lambda: conv.conv_forward(x, wf, ldw, cout, g, bias=b, act=conv.ACT_LRELU, force_tile=t, force_split=1)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f(); f()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(NREP):
                f()
        graphs[t] = gr
    res = {t: [] for t in graphs}
    for rnd in range(5):
        for t, gr in graphs.items():
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[t].append(e0.elapsed_time(e1) / NREP * 1e3)
    out = {'case': name}
    for t, v in res.items():
        us = sorted(v)[len(v) // 2]
        out[NAMES[t]] = {'us': round(us, 1), 'tflops_equiv': round(flops / us / 1e6, 1)}
    print(json.dumps(out), flush=True)
