"""Calibration: what the vendor library (torch.mm -> hipBLASLt / rocBLAS, plain fp32 GEMM, no gather, no epilogue) reaches on the GEMM
shapes of the bench step - the practical ceiling next to the 157.3 TFLOP/s nominal peak.  20 launches per hipGraph, median of 5."""
import torch, json
dev='cuda:0'
for (m,n,k) in [(8192,256,2304),(32768,128,2304),(8192,256,9216),(131072,128,576),(524288,32,288),(2048,512,2304)]:
    a=torch.randn(m,k,device=dev); b=torch.randn(k,n,device=dev)
    for _ in range(3): c=a@b
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): c=a@b
    ts=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)/20)
    t=sorted(ts)[2]
    print(json.dumps({'mnk':[m,n,k],'us':round(t*1e3,1),'tflops':round(2*m*n*k/t/1e9,1)}))
