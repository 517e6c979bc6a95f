"""Headline benchmark: frames/sec of one G+D forward+backward training step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] - fewshot_pose 512x512, per-GPU batch 2, --adaptive_spade
--warp_ref --spade_combine --remove_face_labels --no_flow_gt --no_vgg_loss (SURVEY.md section 8d "C3"), synthetic
labels/images, random-init weights.  One step = the body of the reference's train.py:58-62: D forward (incl. the
no-grad G forward), D backward, Adam(D), G forward, D forward, full backward, Adam(G).  fp32 throughout (exact-fp32
MFMA).  Weak scaling: the per-GPU batch is fixed, gradients are all-reduced over RCCL (flat.py).

Besides the contract fields the JSON line carries
  roofline     - the dominant kernel (the fp32 MFMA implicit-GEMM convolution) priced against the 157.3 TFLOP/s
                 fp32 matrix peak: algorithmic FLOPs per launch / average launch duration, measured with HIP events
                 on the launch stream in an instrumented pass of the same step;
  cpu_baseline - the CPU oracle (oracle/fsv_oracle.py, a port of the reference's algorithm; kind "port") timed on
                 the host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: without this RCCL's cross-process buffer registration fails
# (hipIpcGetMemHandle: invalid argument).  Already exported on the GPU boxes; kept here for bare launches.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"


WITH_VGG = False       # --vgg: add the VGG19 perceptual loss (a "next" row of SURVEY.md 8f; not part of the headline)
WITH_FACE_D = False    # --face-d: BASELINE configs[3] flags (--add_face_D, which needs the VGG loss); not the headline
AMP = 'O0'             # --amp: the reference's apex level string ('O1': fp16 GEMM operands + loss scale; 'bf16x3'); not the headline


def build_opt(size, batch):
    import model_checks as mc
    return mc.make_opt(fineSize=size, loadSize=size, batchSize=batch, warp_ref=True, spade_combine=True,
                       remove_face_labels=True, no_vgg_loss=not (WITH_VGG or WITH_FACE_D), no_flow_gt=True,
                       add_face_D=WITH_FACE_D, amp=AMP)


def make_data(batch, size, seed, device):
    import model_checks as mc
    tl, ti, rl, ri = mc.synth_pose_inputs(batch, size, size, seed)
    tl, ti, rl, ri = [t.to(device) for t in (tl, ti, rl, ri)]
    return [tl, ti, [None, None], [None, None], rl, ri, None, None, None]


def cpu_baseline(size, threads, budget_s=30.0):
    """The CPU oracle (a port of the reference's algorithm) on the host cores: one full iteration (D step + G step,
    both backward passes) at B=1.  The sample is bounded: a 128x128 probe predicts the cost (work scales with the
    pixel count) and the largest of 512 / 256 / 128 that fits the budget is timed; the result is reported as
    `size`x`size`-equivalent frames/s (measured frames/s scaled by the pixel ratio)."""
    import model_checks as mc
    from oracle import fsv_oracle as O
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    M = import_module('few-shot-vid2vid_amd.model')
    torch.set_num_threads(threads)

    def one(sz):
        opt = build_opt(sz, 1)
        model = M.create_model(opt)           # only a source of random-init weights with the right shapes (CPU tensors)
        sdG = {k: v.detach().clone() for k, v in model.netG.state_dict().items()}
        sdD = {k: v.detach().clone() for k, v in model.netD.state_dict().items()}
        del model
        data = mc.synth_pose_inputs(1, sz, sz, 99)
        cfg = O.cfg_from_opt(opt)
        t0 = time.perf_counter()
        mc._oracle_iteration(sdG, sdD, cfg, data, torch.float32)
        return time.perf_counter() - t0
    t128 = one(128)
    t128 = min(t128, one(128))                # first call pays one-time initialisation
    sz = 128
    for cand, factor in ((size, (size / 128.0) ** 2), (256, 4.0)):
        if cand > 128 and t128 * factor <= budget_s:
            sz = cand
            break
    dt = one(sz) if sz != 128 else t128
    equiv = (1.0 / dt) * (sz * sz) / float(size * size)
    return dict(value=round(equiv, 4), unit='frames/s', cores=threads, kind='port',
                sample='1 iteration (D step + G step, fwd+bwd) at B=1, %dx%d, same flags, oracle/fsv_oracle.py: %.1f s; '
                       'reported as %dx%d-equivalent frames/s (x pixel ratio %.3f)' % (sz, sz, dt, size, size, (sz * sz) / float(size * size)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--batch', type=int, default=2, help='per-GPU batch')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--vgg', action='store_true', help='include the VGG19 perceptual loss in the G step')
    ap.add_argument('--face-d', action='store_true', help='config 3: --add_face_D (face discriminator + VGG19 loss)')
    ap.add_argument('--amp', default='O0', help="reference --amp level: O1 = fp16 GEMM operands (fp32 accumulate) + dynamic "
                    "loss scale, bf16x3 = split-bf16 operands; O0 (default, the headline) = exact fp32")
    args = ap.parse_args()
    global WITH_VGG, WITH_FACE_D, AMP
    WITH_VGG = args.vgg
    WITH_FACE_D = args.face_d
    AMP = args.amp

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    group = None
    # FSV_FORCE_DIST=1: exercise the RCCL / bucket / side-stream path in a one-rank group (single-GPU smoke test of
    # the code the driver runs at N > 1; results are identical to the plain path, the step runs eagerly)
    force_dist = os.environ.get('FSV_FORCE_DIST', '0') == '1'
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(backend='nccl', rank=rank, world_size=world)

    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    M = import_module('few-shot-vid2vid_amd.model')
    prof = import_module('few-shot-vid2vid_amd.profile')

    opt = build_opt(args.size, args.batch)
    model = M.create_model(opt).to(device).train()          # identical init on every rank (seed 0), like the reference
    distributed = world > 1 or force_dist
    # N > 1: no autograd hooks, the step is three hipGraph segments with one whole-buffer RCCL all-reduce between them
    segmented = distributed and not args.no_graph
    opt_G, opt_D = model.build_optimizers(world_size=world, process_group=group, force_exchange=force_dist,
                                          overlap=not segmented)
    data = make_data(args.batch, args.size, 1234 + rank, device)

    def backward_of(losses, optimizer):
        loss = sum(torch.mean(x) for x in losses)
        optimizer.zero_grad()
        optimizer.scale_loss(loss).backward()      # identity unless --amp O1
        optimizer.finalize_grads()        # deferred weight-gradient jobs belong to the segment that queued them

    def seg_d():                       # D forward (incl. the no-grad G forward) + D backward
        backward_of(model(data, mode='discriminator'), opt_D)

    def seg_g():                       # Adam(D), then G forward + D forward + full backward
        opt_D.adam()
        backward_of(model(data, mode='generator')[0], opt_G)

    def seg_a():                       # Adam(G)
        opt_G.adam()

    def step():
        if segmented:
            seg_d(); opt_D.exchange_all(); seg_g(); opt_G.exchange_all(); seg_a()
        else:
            d_losses = model(data, mode='discriminator')
            M.loss_backward(opt, d_losses, opt_D, 1)
            g_losses, _, _ = model(data, mode='generator')
            M.loss_backward(opt, g_losses, opt_G, 0)

    use_graph = not args.no_graph
    graphs = None
    n_eager_warm = max(1, min(args.warmup, 2)) if use_graph else args.warmup
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(n_eager_warm):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    mode = 'eager'
    run = step
    if use_graph:
        try:
            if segmented:
                # The RCCL watchdog thread polls its events while a collective is in flight; under the default global
                # capture mode that poll is an illegal call "during capture" and invalidates it.  So: no collective may be
                # pending when a capture starts (synchronize), and the captures only police their own thread.
                graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
                torch.cuda.synchronize()
                with torch.cuda.graph(graphs[0], capture_error_mode='thread_local'):
                    seg_d()
                opt_D.exchange_all()
                torch.cuda.synchronize()
                with torch.cuda.graph(graphs[1], pool=graphs[0].pool(), capture_error_mode='thread_local'):
                    seg_g()
                opt_G.exchange_all()
                torch.cuda.synchronize()
                with torch.cuda.graph(graphs[2], pool=graphs[0].pool(), capture_error_mode='thread_local'):
                    seg_a()

                def run():
                    graphs[0].replay(); opt_D.exchange_all(); graphs[1].replay(); opt_G.exchange_all(); graphs[2].replay()
                mode = 'hipgraph x3 + whole-buffer all-reduce'
            else:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    step()
                run = graph.replay
                mode = 'hipgraph'
        except Exception as e:                      # capture is an optimisation; the step itself is unchanged
            if rank == 0:
                print('graph capture failed (%s); timing the eager step' % str(e).split('\n')[0], file=sys.stderr)
            run = step
            torch.cuda.synchronize()
    for _ in range(max(0, args.warmup - n_eager_warm)):
        run()

    torch.cuda.synchronize()
    if world > 1 or force_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    frames = args.batch * world * args.steps
    result = {
        'metric': 'frames/sec G+D fwd+bwd step, 512x512 fewshot_pose',
        'value': round(frames / elapsed, 4),
        'unit': 'frames/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': {0: 'f32', 1: 'f16 operands / f32 accumulate (--amp, not the headline)', 2: 'bf16x3 operands / f32 accumulate (not the headline)'}[M.amp_mode(opt)],
        'data': 'synthetic',
        'config': {'workload': ('fewshot_pose %dx%d, per-GPU batch %d, adaptive_spade+warp_ref+spade_combine, '
                                'D step + G step (train.py:58-62), Adam included, %sno FlowNet2%s'
                                % (args.size, args.size, args.batch,
                                   ('with VGG19 loss + face discriminator, ' if WITH_FACE_D else 'with VGG19 loss, ')
                                   if (WITH_VGG or WITH_FACE_D) else 'no VGG / ', '' if WITH_FACE_D else ' / face-D')),
                   'global_batch': args.batch * world, 'parallelism': 'dp%d' % world, 'launch': mode,
                   'algorithmic_tflop_per_frame': 1.66},
    }
    rl = None
    if not args.no_roofline:
        # instrumented eager pass: HIP events around every launch of the dominant kernel, on its launch stream.  Every rank
        # runs the step (it contains the gradient exchange), only rank 0 records.
        if rank == 0:
            prof.enable()
        step()
        torch.cuda.synchronize()
        if rank == 0:
            rl = prof.summary()
            prof.disable()
    if rank == 0:
        result['step_tflops'] = round(1.66 * frames / elapsed, 2)
        if rl is not None:
            result['roofline'] = rl['dominant']
            result['kernels'] = rl['by_kernel']
            # HBM bytes per launch of the same kernel from the rocprofv3 PMC passes of this command
            # (profiles/r01_pmc_hbm_traffic.json: FETCH_SIZE doubled per the gfx950 note, + WRITE_SIZE)
            try:
                with open(os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_traffic.json')) as f:
                    pmc = json.load(f)
                want = result['roofline']['kernel'].split('<')[1].split(',V')[0].replace('x', ', ')
                for name, v in pmc.items():
                    if 'fsv_conv_igemm_kernel<' + want in name and name.rstrip().endswith(result['roofline']['kernel'][-2] + '>'):
                        result['roofline']['traffic'] = round((v['read_MB_corrected'] + v['write_MB']) * 1e6)
                        result['roofline']['traffic_unit'] = 'bytes/launch (PMC, profiles/r01_pmc_hbm_traffic.json)'
                        break
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.size, min(os.cpu_count() or 1, 64))
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio (flushed at exit): flush it first so that the JSON line is the
        # last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)


if __name__ == '__main__':
    main()
