"""In-process A/B of the element-wise normalisation kernels (FSV_EW_UNROLL is read at every call): norm_act forward / backward at
the step's large tensors, each variant captured as a graph of 20 launches, interleaved, median of 5.  GB/s = algorithmic bytes."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fsv2v_amd  # noqa
from importlib import import_module
ops = import_module('few-shot-vid2vid_amd.ops')
conv = import_module('few-shot-vid2vid_amd.conv')
dev = torch.device('cuda:0')
NREP = 20
for (n, c, h, w, inst) in [(2, 32, 512, 512, False), (2, 64, 256, 256, False), (2, 256, 64, 64, False), (4, 64, 129, 129, True),
                           (2, 128, 128, 128, False)]:
    x = conv.to_nhwc(torch.randn(n, c, h, w, device=dev)).requires_grad_(True)
    wt, b = torch.ones(c, device=dev, requires_grad=True), torch.zeros(c, device=dev, requires_grad=True)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    dy = conv.to_nhwc(torch.randn(n, c, h, w, device=dev))
    res = {}
    for var in ('0', '1'):
        os.environ['FSV_EW_UNROLL'] = var

        def f():
            y = ops.norm_act(x, wt, b, None if inst else rm, None if inst else rv, instance=inst, eps=1e-5, act=conv.ACT_LRELU)
            y.backward(dy)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f(); f()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(NREP):
                f()
        res[var] = g
    t = {k: [] for k in res}
    for rnd in range(5):
        for k, g in res.items():
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            t[k].append(e0.elapsed_time(e1) / NREP * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    mb = n * c * h * w * 4 / 1e6
    print(json.dumps(dict(case='norm_act fwd+bwd [%d,%d,%d,%d]%s' % (n, c, h, w, ' instance' if inst else ''), tensor_MB=round(mb, 1),
                          grid_stride_us=round(med['0'], 1), unrolled_us=round(med['1'], 1))), flush=True)
