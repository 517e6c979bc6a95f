"""Node census of the captured training step: kernel / memcpy / memset nodes of the hipGraph bench.py replays (hipGraphDebugDotPrint
through torch.cuda.CUDAGraph.debug_dump), memcpy / memset nodes grouped by size - what the runtime's own copy / fill kernels
(__amd_rocclr_copyBuffer / fillBufferAligned in the kernel statistics) are made of.
python tools/graph_nodes.py [--workload pose] [--amp O0] [--dot gpurun_out/step.dot]"""
import argparse
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='pose')
    ap.add_argument('--amp', default='O0')
    ap.add_argument('--dot', default=os.path.join(ROOT, 'gpurun_out', 'step.dot'))
    args = ap.parse_args()
    import bench
    bench.WORKLOAD, bench.AMP = args.workload, args.amp
    wl = bench.WORKLOADS[args.workload]
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    M = import_module('few-shot-vid2vid_amd.model')
    dev = torch.device('cuda:0')
    opt = bench.build_opt(wl['size'], wl['batch'])
    model = M.create_model(opt).to(dev).train()
    opt_G, opt_D = model.build_optimizers()
    data = bench.make_data(wl['batch'], wl['size'], 1234, dev, opt)
    model.early_generator = os.environ.get('FSV_EARLY_G', '1') == '1'

    def step():
        M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
        g_losses, _, _ = model(data, mode='generator')
        M.loss_backward(opt, g_losses, opt_G, 0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        step()
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    graph = ctypes.c_void_p(g.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    rc = hip.hipGraphGetNodes(graph, None, ctypes.byref(n))
    assert rc == 0, rc
    arr = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(graph, arr, ctypes.byref(n)) == 0
    names = {0: 'kernel', 1: 'memcpy', 2: 'memset', 3: 'host', 4: 'graph', 5: 'empty', 6: 'wait_event', 7: 'event_record',
             10: 'mem_alloc', 11: 'mem_free'}

    class MemsetParams(ctypes.Structure):
        _fields_ = [('dst', ctypes.c_void_p), ('elementSize', ctypes.c_uint), ('height', ctypes.c_size_t), ('pitch', ctypes.c_size_t),
                    ('value', ctypes.c_uint), ('width', ctypes.c_size_t)]
    kinds = collections.Counter()
    sized = collections.Counter()
    ne = ctypes.c_size_t(0)
    hip.hipGraphGetEdges(graph, None, None, ctypes.byref(ne))
    for node in arr:
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t))
        kind = names.get(t.value, 'type%d' % t.value)
        kinds[kind] += 1
        if kind == 'memset':
            mp = MemsetParams()
            if hip.hipGraphMemsetNodeGetParams(ctypes.c_void_p(node), ctypes.byref(mp)) == 0:
                sized['memset %d bytes' % (mp.width * mp.elementSize * max(mp.height, 1))] += 1
        elif kind == 'memcpy':
            buf = (ctypes.c_size_t * 32)()
            if hip.hipGraphMemcpyNodeGetParams(ctypes.c_void_p(node), ctypes.byref(buf)) == 0:
                sized['memcpy %d x %d x %d bytes' % (buf[16], buf[17], buf[18])] += 1
    print('nodes:', dict(kinds), 'edges:', ne.value)
    for k, v in sorted(sized.items(), key=lambda kv: -kv[1]):
        print('%4d  %s' % (v, k))


if __name__ == '__main__':
    main()
