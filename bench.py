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
                 on the launch stream around every launch of that kernel in an instrumented eager pass of the step
                 (`avg_launch_us`; the same launches re-issued back to back - warm caches, an upper bound - are reported as
                 `replay_us_warm_cache_upper_bound`); `traffic`: HBM bytes per launch from the committed rocprofv3 PMC passes
                 of this command (profiles/r0N_pmc_hbm_traffic*.json, keyed by kernel, source digest and workload; MB per launch, with
                 `traffic_read_MB` / `traffic_write_MB` beside it), null when no file describes the running build;
                 `frac_in_timed_schedule`: the same kernel timed inside the schedule whose ms_per_step is the headline;
  cpu_baseline - the CPU oracle (oracle/fsv_oracle.py, a port of the reference's algorithm; kind "port": the Python
                 reference cannot travel to the GPU box) timed on the host cores on ONE iteration of the very same
                 workload (512x512, B = 2, same flags; rank 0, N=1 only);
  extras       - (N=1) measurements beside the headline: the north-star target (SPADE-generator forward at 512x512,
                 batch 8, both flag sets, as fractions of the fp32 MFMA peak), the step the reference's shipped
                 script trains (scripts/pose/train_g1.sh: + face discriminator, VGG19 loss, FlowNet2 teacher), and BASELINE
                 configs[4] per rank in its stated arithmetic (`--workload street --amp O1`, run as a child process: value + roofline
                 against the f16 matrix peak).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment (a bare launch) re-executes itself through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`: one rank per GPU either way.

    python bench.py --cpu-baseline-only [--use-reference]    # CPU leg alone (no GPU needed; --use-reference: time the
                                                             # unmodified reference through oracle/ref_import.py instead)
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

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"


WITH_VGG = False       # --vgg: add the VGG19 perceptual loss (a "next" row of SURVEY.md 8f; not part of the headline)
WITH_FACE_D = False    # --face-d: BASELINE configs[3] flags (--add_face_D, which needs the VGG loss); not the headline
AMP = 'O0'             # --amp: the reference's apex level string ('O1': fp16 GEMM operands + loss scale; 'bf16x3'); not the headline


def _synth():
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    return import_module('few-shot-vid2vid_amd.synth')


# --workload: the BASELINE.json configs a single GPU can run.  `pose` (configs[2], C3) is the headline and the
# default; the others print the same line for their own configuration (their `metric` names it) so that C1 / C2 / C5 are
# driver-reproducible next to C3.  tflop = algorithmic work per frame of what the step runs (BASELINE.md section 2:
# D step = G fwd + D fwd + D bwd, G step = G fwd + D fwd + D bwd + G bwd; face256 is defined generator-only).
WORKLOADS = {
    'pose':    dict(size=512, batch=2, tflop=1.66, g_only=False, what='fewshot_pose %dx%d, per-GPU batch %d, adaptive_spade+warp_ref+spade_combine'),
    'face128': dict(size=128, batch=1, tflop=0.0655, g_only=False, what='fewshot_face %dx%d, per-GPU batch %d, adaptive_spade (BASELINE configs[0] on the GPU)'),
    'face256': dict(size=256, batch=4, tflop=0.1407, g_only=True, what='fewshot_face %dx%d, per-GPU batch %d, adaptive_spade, generator forward + backward only (BASELINE configs[1])'),
    'street':  dict(size=1024, batch=1, tflop=1.7845, g_only=False, what='fewshot_street %dx%d (W x H = 1024x512), label_nc 35, per-GPU batch %d, adaptive_spade (BASELINE configs[4] per rank)'),
}
WORKLOAD = 'pose'


def build_opt(size, batch, vgg=None, face_d=None, flow_gt=False, workload=None):
    vgg = WITH_VGG if vgg is None else vgg
    face_d = WITH_FACE_D if face_d is None else face_d
    workload = WORKLOAD if workload is None else workload
    if workload in ('face128', 'face256'):
        return _synth().make_opt(dataset_mode='fewshot_face', input_nc=1, fineSize=size, loadSize=size, batchSize=batch,
                                 no_vgg_loss=not vgg, amp=AMP)
    if workload == 'street':      # data/fewshot_street_dataset.py:19-27: aspect_ratio 2 (W = fineSize, H = fineSize / 2), one-hot labels
        return _synth().make_opt(dataset_mode='fewshot_street', label_nc=35, input_nc=3, aspect_ratio=2.0, fineSize=size,
                                 loadSize=size, batchSize=batch, no_vgg_loss=not vgg, amp=AMP)
    return _synth().make_opt(fineSize=size, loadSize=size, batchSize=batch, warp_ref=True, spade_combine=True,
                             remove_face_labels=True, no_vgg_loss=not (vgg or face_d), no_flow_gt=not flow_gt,
                             add_face_D=face_d, amp=AMP)


def synth_inputs(opt, batch, seed):
    h, w = int(opt.fineSize / opt.aspect_ratio), opt.fineSize
    if opt.label_nc != 0:
        return _synth().synth_street_inputs(batch, h, w, seed, opt.label_nc)
    return _synth().synth_pose_inputs(batch, h, w, seed, opt.input_nc)


def make_data(batch, size, seed, device, opt=None):
    tl, ti, rl, ri = _synth().synth_pose_inputs(batch, size, size, seed) if opt is None else synth_inputs(opt, batch, seed)
    tl, ti, rl, ri = [t.to(device) for t in (tl, ti, rl, ri)]
    return [tl, ti, [None, None], [None, None], rl, ri, None, None, None]


def cpu_baseline(size, batch, threads, use_reference=False):
    """One full iteration (D step + G step, both backward passes - the body of train.py:58-62) of the bench workload on
    the host cores, at the bench configuration itself (no scaling).  kind "port": oracle/fsv_oracle.py, the CPU
    restatement of the reference's algorithm (pinned to the reference by tests/golden/); kind "reference": the
    unmodified reference modules imported from /root/reference through oracle/ref_import.py (build container only -
    the Python reference does not exist on the GPU box).  A 64x64 iteration first pays the one-time initialisation."""
    syn = _synth()
    torch.set_num_threads(threads)
    if use_reference:
        from oracle import ref_import
        ref_import.install_shims()
        from models.loss_collector import loss_backward

        def one(sz, b):
            flags = ('--dataset_mode fewshot_pose --aspect_ratio 1 --fineSize %d --loadSize %d --adaptive_spade --warp_ref '
                     '--spade_combine --remove_face_labels --no_flow_gt --no_vgg_loss --gpu_ids -1 --batchSize %d' % (sz, sz, b))
            opt, model = ref_import.build_model(flags.split())
            tl, ti, rl, ri = syn.synth_pose_inputs(b, sz, sz, 99)
            data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
            t0 = time.perf_counter()
            loss_backward(opt, model(data, mode='discriminator'), model.optimizer_D, 1)
            g_losses, _, _ = model(data, mode='generator')
            loss_backward(opt, g_losses, model.optimizer_G, 0)
            return time.perf_counter() - t0
        kind, what = 'reference', 'unmodified reference modules (oracle/ref_import.py)'
    else:
        from oracle import fsv_oracle as O
        from importlib import import_module
        M = import_module('few-shot-vid2vid_amd.model')

        def one(sz, b):
            opt = build_opt(sz, b, vgg=False, face_d=False)
            opt.amp = 'O0'                        # the CPU leg is the reference's fp32 arithmetic whatever the GPU leg runs
            model = M.create_model(opt)           # only a source of random-init weights with the right shapes (CPU tensors)
            sdG = {k: v.detach().clone() for k, v in model.netG.state_dict().items()}
            sdD = {k: v.detach().clone() for k, v in model.netD.state_dict().items()}
            del model
            data = synth_inputs(opt, b, 99)
            cfg = O.cfg_from_opt(opt)
            t0 = time.perf_counter()
            O.iteration(sdG, sdD, cfg, data, torch.float32)
            return time.perf_counter() - t0
        kind, what = 'port', 'oracle/fsv_oracle.py'
    if WORKLOAD != 'pose' and use_reference:
        raise SystemExit("--use-reference times the headline workload only")
    one(64, 1)
    dt = one(size, batch)
    return dict(value=round(batch / dt, 4), unit='frames/s', cores=threads, kind=kind,
                sample='1 iteration (D step + G step, fwd+bwd, train.py:58-62) of the bench workload itself: %dx%d, B=%d, '
                       'same flags, %s: %.1f s' % (size, size, batch, what, dt))


def pmc_traffic(label):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this command (FETCH_SIZE x 2 per
    the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE, tools/pmc_traffic.py) - only from a file that was recorded for the very
    kernel sources this library was built from (source digest) and for this workload; None otherwise: counters cannot be
    collected in-run."""
    import glob
    try:
        from importlib import import_module
        build = import_module('few-shot-vid2vid_amd.build')
        digest = build.source_digest()
        # label 'fsv_conv_igemm_kernel<64x128lds,V4>' -> PMC key 'void fsv_conv_igemm_kernel<64, 128, ...'
        name, dims = label.split('<')[0], label.split('<')[1].split(',')[0]
        dims = ''.join(ch if (ch.isdigit() or ch == 'x') else ' ' for ch in dims).split()[0].split('x')
        for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic*.json')), reverse=True):
            with open(path) as f:
                d = json.load(f)
            b = d.get('_build', {})
            if b.get('source_digest') != digest or b.get('workload', 'pose') != WORKLOAD or b.get('amp', 'O0') != AMP:
                continue
            for k, v in d.items():
                if k.startswith('void %s<%s, %s,' % (name, dims[0], dims[1])):
                    return dict(read_MB=v['read_MB_corrected'], write_MB=v['write_MB'],
                                total_MB=round(v['read_MB_corrected'] + v['write_MB'], 2), unit='MB per launch',
                                source='profiles/' + os.path.basename(path), commit=b.get('commit'))
    except Exception:
        return None
    return None


def _time_graph(fn, n=5):
    """ms per call of fn() replayed as a hipGraph (eager warm-up on a side stream first)"""
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def extras(device, size, steps=5):
    """Measurements beside the headline (rank 0, N=1): see the module docstring.  Each in its own try: a failure is
    reported as a string, the headline line is never lost."""
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    syn = _synth()
    net = import_module('few-shot-vid2vid_amd.networks')
    M = import_module('few-shot-vid2vid_amd.model')
    out = {}
    # ---- north-star target: SPADE-generator forward, 512x512, batch 8, training-mode statistics, no autograd tape ----------
    for name, combine, gflop in (('g_forward_bs8_warp_combine', True, 386.6), ('g_forward_bs8_adaptive_spade_only', False, 179.2)):
        try:
            opt = syn.make_opt(fineSize=size, loadSize=size, warp_ref=combine, spade_combine=combine)
            torch.manual_seed(0)
            G = net.define_G(opt).to(device).train()
            tl, ti, rl, ri = syn.synth_pose_inputs(8, size, size, 1)
            label, rl, ri = tl[:, 0].to(device), rl.to(device), ri.to(device)

            def fwd():
                with torch.no_grad():
                    return G(label, rl, ri, [None, None])
            ms = _time_graph(fwd, steps)
            tf = gflop * 8 * (size / 512.0) ** 2 / ms
            # what the device executes: the flops of every MFMA launch of one eager pass, counted by the launch wrappers
            prof = import_module('few-shot-vid2vid_amd.profile')
            prof.enable()
            fwd()
            by = prof.summary()['by_kernel']
            prof.disable()
            ex_gflop = sum(v['gflop_per_launch'] * v['launches'] for v in by.values())
            out[name] = dict(ms=round(ms, 2), gflop_per_frame=gflop, tflops=round(tf, 2), peak_tflops=FP32_MFMA_PEAK_TFLOPS,
                             frac_fp32_mfma_peak=round(tf / FP32_MFMA_PEAK_TFLOPS, 4), frames_per_s=round(8e3 / ms, 2),
                             executed_gflop_per_frame=round(ex_gflop / 8, 1), tflops_executed=round(ex_gflop / ms, 2),
                             frac_fp32_mfma_peak_executed=round(ex_gflop / ms / FP32_MFMA_PEAK_TFLOPS, 4),
                             arithmetic="gflop_per_frame counts the REFERENCE's multiply-adds (nn.Upsample(2) -> conv3x3 as 36 "
                                        "products per source pixel and channel pair); the device executes 16 of them in the large "
                                        "up-sampling layers (DESIGN.md 4d): tflops / frac_fp32_mfma_peak are the rate of the "
                                        "reference's work, *_executed the matrix pipe's utilisation")
            del G, fwd
        except Exception as e:          # noqa: BLE001
            out[name] = 'failed: %s' % str(e).split('\n')[0][:200]
        torch.cuda.empty_cache()
    # ---- the step scripts/pose/train_g1.sh trains: + face discriminator, VGG19 loss, FlowNet2 teacher (random weights) ------
    try:
        fn = import_module('few-shot-vid2vid_amd.flownet2')
        opt = build_opt(size, 2, vgg=True, face_d=True, flow_gt=True)
        model = M.create_model(opt).to(device).train()
        opt_G, opt_D = model.build_optimizers()
        teacher = fn.FlowNet(opt).to(device).eval()
        data = make_data(2, size, 4321, device)

        def full_step():
            with torch.no_grad():           # train.py:44-48: teacher flow / confidence of (reference, target), no prev frame
                flow_gt, conf_gt = teacher([data[1], data[5]], 0)
            d = list(data)
            d[2], d[3] = flow_gt, conf_gt
            M.loss_backward(opt, model(d, mode='discriminator'), opt_D, 1)
            g_losses, _, _ = model(d, mode='generator')
            M.loss_backward(opt, g_losses, opt_G, 0)
        ms = _time_graph(full_step, steps)
        out['shipped_step_face_d_vgg_flownet2'] = dict(ms_per_step=round(ms, 2), frames_per_s=round(2e3 / ms, 2), batch=2,
                                                       note='scripts/pose/train_g1.sh flags at 512x512: --add_face_D, VGG19 '
                                                            'loss, FlowNet2 teacher forward (162.5 M random weights) inside '
                                                            'the timed step')
    except Exception as e:              # noqa: BLE001
        out['shipped_step_face_d_vgg_flownet2'] = 'failed: %s' % str(e).split('\n')[0][:200]
    torch.cuda.empty_cache()
    # ---- BASELINE configs[4] per rank in its stated arithmetic: street 1024x512, label_nc 35, `--amp O1` (fp16 MFMA path) -------
    # the same script in a child process (the operand arithmetic is process-wide, like apex's amp.initialize): its own line,
    # value + roofline against the f16 peak, so that the driver's default command records this configuration too
    try:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'street', '--amp', 'O1', '--steps', '10', '--warmup', '3',
               '--no-cpu-baseline', '--no-extras']
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        lines = [ln for ln in pr.stdout.splitlines() if ln.startswith('{')]
        if pr.returncode != 0 or not lines:
            raise RuntimeError('exit %d: %s' % (pr.returncode, (pr.stderr or pr.stdout).strip().split('\n')[-1][:160]))
        d = json.loads(lines[-1])
        keep = {k: d.get(k) for k in ('metric', 'value', 'unit', 'ms_per_step', 'dtype', 'steps', 'warmup', 'step_tflops',
                                      'step_tflops_executed')}
        keep['config'] = d.get('config')
        rf = d.get('roofline') or {}
        keep['roofline'] = {k: rf.get(k) for k in ('kernel', 'bound', 'launches', 'gflop_per_launch', 'avg_launch_us', 'achieved', 'peak',
                                                   'unit', 'frac', 'in_timed_schedule', 'frac_in_timed_schedule', 'traffic', 'traffic_read_MB',
                                                   'traffic_write_MB') if k in rf}
        out['street_1024x512_nc35_amp_O1'] = keep
    except Exception as e:              # noqa: BLE001
        out['street_1024x512_nc35_amp_O1'] = 'failed: %s' % str(e).split('\n')[0][:200]
    return out


def spawn_ranks(n):
    """Bare `python bench.py --gpus N` (N > 1, no launcher environment): re-execute this command line as N ranks of one node
    through torch.distributed.run, exactly the command the driver uses; rank 0 of the children prints the JSON line."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='pose', choices=sorted(WORKLOADS),
                    help='pose = BASELINE configs[2] (the headline, default); face128 / face256 / street = configs[0] / [1] / [4] '
                         'per rank on one GPU, each with its own metric string')
    ap.add_argument('--size', type=int, default=None, help='fineSize (default: the workload\'s)')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the workload\'s)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true', help='run only the CPU leg (no GPU needed)')
    ap.add_argument('--use-reference', action='store_true', help='CPU leg: time the unmodified reference (build container)')
    ap.add_argument('--vgg', action='store_true', help='include the VGG19 perceptual loss in the G step')
    ap.add_argument('--face-d', action='store_true', help='config 3: --add_face_D (face discriminator + VGG19 loss)')
    ap.add_argument('--amp', default='O0', help="reference --amp level: O1 = fp16 GEMM operands (fp32 accumulate) + dynamic "
                    "loss scale, bf16x3 = split-bf16 operands; O0 (default, the headline) = exact fp32")
    ap.add_argument('--ngf', type=int, default=32, help='network width (32 = the reference default = the headline; smaller '
                    'values only for the emulated launcher test)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and not args.cpu_baseline_only:
        raise SystemExit(spawn_ranks(args.gpus))
    global WITH_VGG, WITH_FACE_D, AMP, WORKLOAD
    WITH_VGG = args.vgg
    WITH_FACE_D = args.face_d
    AMP = args.amp
    WORKLOAD = args.workload
    wl = WORKLOADS[WORKLOAD]
    if args.size is None:
        args.size = wl['size']
    if args.batch is None:
        args.batch = wl['batch']
    if WORKLOAD != 'pose' and WITH_FACE_D:
        raise SystemExit("--face-d belongs to the pose workload")

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.size, args.batch, min(os.cpu_count() or 1, 64), args.use_reference)), flush=True)
        return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # FSV2V_EMU=1 (CPU test-suite only): the SIMT-emulated build of the kernel library on host tensors, gloo instead of
    # RCCL, eager step - exercises the launcher / rank / JSON plumbing of this script without a GPU (tests/test_tools.py);
    # the line it prints says so and is never a measurement
    emulated = os.environ.get('FSV2V_EMU', '0') == '1'
    if emulated:
        device = torch.device('cpu')
        args.no_graph = args.no_roofline = args.no_cpu_baseline = args.no_extras = True
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        if world > torch.cuda.device_count():
            raise SystemExit("bench.py: %d ranks but %d visible GPUs (one rank per GPU)" % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    sync = (lambda: None) if emulated else torch.cuda.synchronize
    group = None
    # FSV_FORCE_DIST=1: exercise the RCCL / bucket / side-stream path in a one-rank group (single-GPU smoke test of
    # the code the driver runs at N > 1; results are identical to the plain path, the step runs eagerly)
    force_dist = os.environ.get('FSV_FORCE_DIST', '0') == '1'
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        import datetime
        # a CUDA/HIP error inside the process group's watchdog thread (the event query that can collide with a graph capture,
        # graph_step._quiesce) is logged instead of re-thrown: it must not take the benchmark process down - the step falls back
        # to the eager exchange below and the JSON line says so
        os.environ.setdefault('TORCH_NCCL_RETHROW_CUDA_ERRORS', '0')
        if emulated:
            dist.init_process_group(backend='gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend='nccl', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180),
                                    device_id=device)

    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    M = import_module('few-shot-vid2vid_amd.model')
    prof = import_module('few-shot-vid2vid_amd.profile')

    opt = build_opt(args.size, args.batch)
    opt.ngf = opt.ndf = opt.nff = args.ngf
    model = M.create_model(opt).to(device).train()          # identical init on every rank (seed 0), like the reference
    distributed = world > 1 or force_dist
    # N > 1: no autograd hooks; the step runs as hipGraph segments (graph_step.GraphedIteration) - D | Adam(D) + G forward +
    # first piece of the G backward | rest of the G backward | Adam(G) - with the RCCL all-reduces between them; the
    # exchange of the decoder-stage gradients (51 % of the generator's parameters) runs on a side stream next to the third
    # graph.  No collective is captured.
    segmented = distributed and (emulated or not args.no_graph)
    g_only = wl['g_only']
    if g_only and distributed:
        raise SystemExit("--workload %s is a single-GPU generator-only measurement" % WORKLOAD)

    # pieces of the generator's backward pass: 2 (the stage boundary in front of the decoder) or, FSV_BENCH_PIECES=3, a second
    # boundary behind the reference encoders (the middle range of the gradient exchange then overlaps the third piece too)
    _PIECES = 3 if os.environ.get('FSV_BENCH_PIECES', '2') == '3' else True

    def build_optimizers(seg):
        return model.build_optimizers(world_size=world, process_group=group, force_exchange=force_dist,
                                      overlap=not seg, split_backward=(_PIECES if (seg or (not g_only and os.environ.get('FSV_BENCH_SPLIT', '1') == '1')) else False))
    opt_G, opt_D = build_optimizers(segmented)
    data = make_data(args.batch, args.size, 1234 + rank, device, opt)

    # one GPU: the generator-mode forward pass starts on a side stream next to the discriminator step (model.early_generator;
    # FSV_EARLY_G=0: the sequential order, in-box A/B).  With a process group the exchange sits between the two steps.
    model.early_generator = (not distributed) and os.environ.get('FSV_EARLY_G', '1') == '1'

    def step():
        d_losses = model(data, mode='discriminator')
        M.loss_backward(opt, d_losses, opt_D, 1)
        g_losses, _, _ = model(data, mode='generator')
        M.loss_backward(opt, g_losses, opt_G, 0)

    if g_only:
        # BASELINE configs[1]: "G-only fwd/bwd": the generator called directly, the mean of its image as the scalar, Adam(G) included
        conv_mod = import_module('few-shot-vid2vid_amd.conv')
        label, ref_label, ref_img = data[0][:, 0], data[4], data[5]

        def step():              # noqa: F811
            with conv_mod.stats_pass(device):
                img = model.netG(label, ref_label, ref_img, [None, None])[0]
            M.loss_backward(opt, [img.mean().view(1, 1)], opt_G, 0)

    use_graph = not args.no_graph
    mode = 'eager'
    run = step
    n_eager_warm = 0
    if segmented:
        # N > 1: hipGraph segments with the RCCL all-reduces between them.  Nothing in here may cost the scaling line: if
        # building, warming or capturing the segmented iteration fails, the optimisers are re-laid for the eager overlapped
        # exchange (bucket hooks on a side stream, flat.py) and the step runs eagerly - `config.launch` says which one ran.
        try:
            gs = import_module('few-shot-vid2vid_amd.graph_step')
            n_eager_warm = max(1, min(args.warmup, 2))
            gi = gs.GraphedIteration(model, opt, warmup=n_eager_warm)
            inject = os.environ.get('FSV_BENCH_INJECT_CAPTURE_FAILURE', '')       # tests/test_tools.py
            if inject == '1':
                gi._can_capture = True

                def _boom(e, save_images):
                    raise RuntimeError("injected capture failure (hipErrorCapturedEvent stand-in)")
                gi._capture = _boom
            if inject == '2':
                raise RuntimeError("injected GraphedIteration construction failure")

            def run():               # noqa: F811
                gi(data)
            for _ in range(n_eager_warm + 1):          # eager warm-up calls, then the capture (followed by its first replay)
                run()
            n_eager_warm += 1
            mode = gi.launch_mode()
            if emulated:
                mode += ' + gloo (CPU test infrastructure, not a measurement)'
        except Exception as e:                          # noqa: BLE001
            if rank == 0:
                print('segmented graph step failed (%s); eager step with the overlapped bucket exchange' % str(e).split('\n')[0],
                      file=sys.stderr)
            sync()
            segmented = False
            import_module('few-shot-vid2vid_amd.networks').BackwardCut.abandon_all()
            opt_G, opt_D = build_optimizers(False)
            run = step
            n_eager_warm = 0
            mode = 'eager fallback (segmented graph step failed: %s), overlapped RCCL bucket exchange' % str(e).split('\n')[0][:160]
            if emulated:
                mode += ' + gloo, emulated kernels (CPU test infrastructure, not a measurement)'
    elif emulated:
        n_eager_warm = 0
        mode = 'emulated kernels on host tensors + gloo (CPU test infrastructure, not a measurement)'
    else:
        n_eager_warm = max(1, min(args.warmup, 2)) if use_graph else args.warmup
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(n_eager_warm):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if use_graph:
            try:
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

    # N > 1: every collective of the timed steps is logged (bytes, issue -> complete time on its stream) so that a scaling
    # record explains itself: `exchange` in the JSON line
    xlog = None
    if distributed:
        xlog = []
        opt_G.exchange_log = opt_D.exchange_log = xlog
    sync()
    if world > 1 or force_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    frames = args.batch * world * args.steps
    hh, ww = int(opt.fineSize / opt.aspect_ratio), opt.fineSize
    if WORKLOAD == 'pose':
        metric = 'frames/sec G+D fwd+bwd step, 512x512 fewshot_pose'
        what = (wl['what'] % (args.size, args.size, args.batch) +
                ', D step + G step (train.py:58-62), Adam included, %sno FlowNet2%s'
                % (('with VGG19 loss + face discriminator, ' if WITH_FACE_D else 'with VGG19 loss, ')
                   if (WITH_VGG or WITH_FACE_D) else 'no VGG / ', '' if WITH_FACE_D else ' / face-D'))
    else:
        metric = ('frames/sec %s, %dx%d %s' % ('G fwd+bwd' if g_only else 'G+D fwd+bwd step', ww, hh, opt.dataset_mode))
        what = (wl['what'] % (ww, hh, args.batch) + (', Adam(G) included' if g_only else
                                                     ', D step + G step (train.py:58-62), Adam included, %s'
                                                     % ('with VGG19 loss' if WITH_VGG else 'no VGG')))
    tflop_per_frame = wl['tflop'] * (args.size / float(wl['size'])) ** 2
    result = {
        'metric': metric,
        'value': round(frames / elapsed, 4),
        'unit': 'frames/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': {0: 'f32', 1: 'f16', 2: 'bf16x3 operands / f32 accumulate (not the headline)'}[M.amp_mode(opt)],
        'data': 'synthetic',
        'config': {'workload': what,
                   'global_batch': args.batch * world, 'parallelism': 'dp%d' % world, 'launch': mode,
                   'algorithmic_tflop_per_frame': round(tflop_per_frame, 4)},
    }
    if getattr(model, 'early_generator', False) and not g_only:
        result['config']['schedule'] = ('discriminator step (forward, backward, Adam) on a side stream next to the generator-mode forward '
                                        'pass: parallel branches of the captured graph, same kernels and per-network order')
    if M.amp_mode(opt) == 1:
        result['config']['arithmetic'] = ('--amp O1: GEMM operands and activations between the narrow layers in IEEE half, fp32 '
                                          'accumulation (v_mfma_f32_32x32x16_f16), dynamic loss scale; not the fp32 headline')
    rl = None
    if not args.no_roofline:
        # instrumented eager pass: HIP events around every launch of the MFMA kernels, on their launch stream; the dominant
        # kernel's launches are then re-issued back to back under one event pair (profile.summary: an eager pass idles
        # between launches and reads ~12 % long).  Every rank runs the step (it contains the gradient exchange), only rank 0
        # records.
        streams = import_module('few-shot-vid2vid_amd.streams')
        forked, streams.ENABLED = streams.ENABLED, False        # one stream: the brackets / the replay see every launch
        if rank == 0:
            prof.enable()
        step()
        torch.cuda.synchronize()
        streams.ENABLED = forked
        if rank == 0:
            rl = prof.summary()
            prof.disable()
        # second instrumented pass: the dominant kernel's launches alone between event pairs.  The first pass brackets ~450
        # launches and the eager step idles between them (its figure reads ~10 % long against rocprofv3's in-graph duration of the
        # same kernel); with ~130 pairs the step runs close to its uninstrumented pace.  avg_launch_us / achieved / frac are this
        # pass, bracketed_us stays the first one.
        dom = rl['dominant']['kernel'] if (rl is not None and rl.get('dominant')) else None
        streams.ENABLED = False
        if rank == 0 and dom is not None:
            prof.enable(only=dom)
        step()
        torch.cuda.synchronize()
        streams.ENABLED = forked
        if rank == 0 and dom is not None:
            n2, t2 = prof.bracket_average(dom)
            prof.disable()
            d = rl['dominant']
            if n2 == d['launches'] and t2 > 0:
                d['avg_launch_us'] = round(t2 * 1e6, 2)
                d['achieved'] = round(d['gflop_per_launch'] * 1e9 / t2 / 1e12, 2)
                d['frac'] = round(d['achieved'] / d['peak'], 4)
        # third pass (one GPU, graph mode): the step captured ONCE MORE with device-side time stamps around the dominant kernel's
        # launches (csrc/stamp.hip: HIP events cannot go into a replayable graph, a kernel that writes the GPU's wall clock can)
        # and replayed - the launches timed where they sit in the REPLAYED iteration, with its neighbours, clocks and caches;
        # an empty stamp pair per launch measures the bracket's own cost.  When it works, avg_launch_us / achieved / frac are
        # priced on it and the eager bracket of the second pass is kept as eager_bracket_us.
        if rank == 0 and dom is not None and mode == 'hipgraph' and os.environ.get('FSV_BENCH_STAMPS', '1') == '1':
            try:
                d = rl['dominant']
                prof.enable(only=dom, events=False)
                prof.stamp_begin(dom, int(d['launches']) + 8, device)
                # ONE stream for this capture: with the side branches on, a launch's wall time includes the kernels of the other
                # streams it shares the chip with (113 us against 92 us for the same kernel, in-box) - that is the schedule's
                # business, not the kernel's
                streams.ENABLED = os.environ.get('FSV_BENCH_STAMP_STREAMS', '0') == '1'
                try:
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2):
                        step()
                finally:
                    streams.ENABLED = forked
                st = prof.stamp_end()
                prof.disable()
                for _ in range(3):
                    g2.replay()
                n3, full, empty = prof.stamp_result(st)
                if n3 == d['launches'] and full > empty > 0:
                    t3 = full - empty
                    d['eager_bracket_us'] = d['avg_launch_us']
                    d['avg_launch_us'] = round(t3 * 1e6, 2)
                    d['stamp_pair_us'] = round(empty * 1e6, 2)
                    d['achieved'] = round(d['gflop_per_launch'] * 1e9 / t3 / 1e12, 2)
                    d['frac'] = round(d['achieved'] / d['peak'], 4)
                    d['timing'] = ('device-side time stamps (wall clock written by one-work-item kernels, csrc/stamp.hip) around every '
                                   'launch of this kernel INSIDE a replayed hipGraph of the step captured on ONE stream, minus the cost '
                                   'of an empty stamp pair (stamp_pair_us); in_timed_schedule: the same stamps in a capture with the side '
                                   'streams ON - the schedule whose ms_per_step is the headline -, where the bracket of a launch also '
                                   'holds the time its workgroups wait for / share the chip with the other branches\' kernels '
                                   '(rocprofv3\'s per-dispatch duration in that schedule lies between the two: profiles/); '
                                   'eager_bracket_us / bracketed_us: HIP events around the launches in instrumented '
                                   'eager passes (this kernel alone / every MFMA kernel bracketed); '
                                   'replay_us_warm_cache_upper_bound: the same launches re-issued back to back')
                del g2
                # ... and the same stamped capture in the schedule that was TIMED (side streams on; round-4 review item 7): both are
                # reported, `frac` stays the one-stream figure (the kernel), `in_timed_schedule.frac` is the kernel + its neighbours
                if forked and n3 == d['launches'] and full > empty > 0:
                    prof.enable(only=dom, events=False)
                    prof.stamp_begin(dom, int(d['launches']) + 8, device)
                    try:
                        g3 = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g3):
                            step()
                    finally:
                        st = prof.stamp_end()
                        prof.disable()
                    for _ in range(3):
                        g3.replay()
                    n4, full4, empty4 = prof.stamp_result(st)
                    if n4 == d['launches'] and full4 > empty4 > 0:
                        t4 = full4 - empty4
                        a4 = d['gflop_per_launch'] * 1e9 / t4 / 1e12
                        d['in_timed_schedule'] = dict(avg_launch_us=round(t4 * 1e6, 2), achieved=round(a4, 2),
                                                      frac=round(a4 / d['peak'], 4), stamp_pair_us=round(empty4 * 1e6, 2))
                    del g3
            except Exception as e:                      # noqa: BLE001 - the measurement must never cost the bench line
                prof.stamp_end(); prof.disable()
                print('stamped capture failed (%s); roofline priced on the eager brackets' % str(e).split('\n')[0], file=sys.stderr)
                torch.cuda.synchronize()
    if xlog is not None:
        opt_G.exchange_log = opt_D.exchange_log = None
        if rank == 0:
            from importlib import import_module as _im
            ex = _im('few-shot-vid2vid_amd.flat').FlatAdam.exchange_summary(xlog)
            result['exchange'] = ex
            result['exchange_note'] = ('gradient all-reduces per step (RCCL; sum, the 1/world factor sits in the Adam kernel): bytes and '
                                       'average issue -> complete ms of each; side_stream: runs next to the following graph segment. '
                                       'ranks %d; critical-path collective: the last generator range' % world)
    if rank == 0:
        result['step_tflops'] = round(tflop_per_frame * frames / elapsed, 2)
        if rl is not None:
            result['roofline'] = rl['dominant']
            result['kernels'] = rl['by_kernel']
            # step_tflops prices the REFERENCE's multiply-adds; the device executes fewer:
            # the flops of every MFMA launch of the instrumented pass, counted by the launch wrappers, over the same time
            ex_gflop = sum(v['gflop_per_launch'] * v['launches'] for v in rl['by_kernel'].values())
            result['step_gflop_executed'] = round(ex_gflop, 1)
            result['step_tflops_executed'] = round(ex_gflop * 1e-3 * world / (elapsed / args.steps), 2)
            # scalar twins of the nested objects (a consumer that keeps scalar fields only still sees the in-schedule figure
            # and the PMC traffic): `traffic` = HBM MB per launch (read + written), null without a PMC file of this build
            rf = result['roofline']
            tr = pmc_traffic(rf['kernel'])
            rf['traffic'] = None if tr is None else tr['total_MB']
            rf['traffic_unit'] = 'MB per launch (HBM, rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE)'
            rf['traffic_read_MB'] = None if tr is None else tr['read_MB']
            rf['traffic_write_MB'] = None if tr is None else tr['write_MB']
            rf['traffic_source'] = None if tr is None else '%s @ %s' % (tr['source'], tr.get('commit'))
            its = rf.get('in_timed_schedule')
            rf['frac_in_timed_schedule'] = None if not its else its['frac']
            rf['avg_launch_us_in_timed_schedule'] = None if not its else its['avg_launch_us']
        if world == 1 and not args.no_cpu_baseline and not g_only:
            result['cpu_baseline'] = cpu_baseline(args.size, args.batch, min(os.cpu_count() or 1, 64))
        if world == 1 and not args.no_extras and not (WITH_VGG or WITH_FACE_D) and AMP == 'O0' and WORKLOAD == 'pose':
            result['extras'] = extras(device, args.size)
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
