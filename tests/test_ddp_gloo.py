"""world_size-2 data-parallel test on CPU (gloo): the flat-buffer gradient exchange keeps replicas in lock-step and
averages gradients exactly like a single process seeing both shards (BatchNorm statistics are per replica by design)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir, overlap=True, iters=1):
    os.environ['FSV2V_EMU'] = '1'
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    torch.set_num_threads(1)
    import model_checks as mc
    dist.init_process_group('gloo', rank=rank, world_size=world)
    M = mc._model()
    opt = mc.tiny_opt(ngf=4, ndf=4, dataset_mode='fewshot_face', input_nc=1, fineSize=32, loadSize=32, n_downsample_G=3,
                      n_adaptive_layers=2)
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    model.train()
    opt_G, opt_D = model.build_optimizers(world_size=world, overlap=overlap)
    assert (opt_G.finalizer is not None) == (not overlap)       # deferred weight gradients only without autograd hooks
    tl, ti, rl, ri = mc.synth_pose_inputs(1, 32, 32, 100 + rank, 1)
    data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
    for it in range(iters):
        d = model(data, mode='discriminator'); M.loss_backward(opt, d, opt_D, 1)
        if it == 0:
            gD = opt_D.flat_g.clone()
        g, _, _ = model(data, mode='generator'); M.loss_backward(opt, g, opt_G, 0)
    torch.save(dict(gD=gD, gG=opt_G.flat_g.clone(), pD=opt_D.flat_p.clone(), pG=opt_G.flat_p.clone(),
                    nb=len(opt_G.buckets)), os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.destroy_process_group()


def _single(rank_seed, out):
    """the same step in one process without any exchange: per-rank local gradients"""
    os.environ['FSV2V_EMU'] = '1'
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import model_checks as mc
    M = mc._model()
    opt = mc.tiny_opt(ngf=4, ndf=4, dataset_mode='fewshot_face', input_nc=1, fineSize=32, loadSize=32, n_downsample_G=3,
                      n_adaptive_layers=2)
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    model.train()
    opt_G, opt_D = model.build_optimizers(world_size=1)
    opt_G.set_lr(0.0); opt_D.set_lr(0.0)
    tl, ti, rl, ri = mc.synth_pose_inputs(1, 32, 32, 100 + rank_seed, 1)
    data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
    d = model(data, mode='discriminator'); M.loss_backward(opt, d, opt_D, 1)
    gD = opt_D.flat_g.clone()
    return gD


def test_two_rank_gradient_exchange(emu_lib, tmp_path):
    world, port = 2, 29611
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    # after the exchange both ranks hold the same (summed) gradients and therefore the same parameters
    assert torch.equal(r0['gD'], r1['gD']) and torch.equal(r0['gG'], r1['gG'])
    assert torch.equal(r0['pD'], r1['pD']) and torch.equal(r0['pG'], r1['pG'])
    # ... and the sum equals the two local gradients added (D step: weights identical on both ranks at that point)
    local = _single(0, None) + _single(1, None)
    scale = float(local.abs().max())
    assert float((r0['gD'] - local).abs().max()) <= 1e-5 * scale


def test_two_rank_whole_buffer_exchange(emu_lib, tmp_path):
    """the mode bench.py uses at N > 1 (overlap=False): kernel-side gradient sinks, deferred weight-gradient finalisation
    and the detached small-parameter gradients (second iteration) all land in the flat buffer before ONE all-reduce"""
    world, port = 2, 29613
    mp.spawn(_worker, args=(world, port, str(tmp_path), False, 2), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    assert r0['nb'] == 0
    assert torch.equal(r0['gD'], r1['gD']) and torch.equal(r0['gG'], r1['gG'])
    assert torch.equal(r0['pD'], r1['pD']) and torch.equal(r0['pG'], r1['pG'])
    local = _single(0, None) + _single(1, None)
    scale = float(local.abs().max())
    assert float((r0['gD'] - local).abs().max()) <= 1e-5 * scale


def _split_worker(rank, world, port, out_dir, split, serial=False, split_adam=False):
    """the N > 1 bench path: GraphedIteration over hook-free optimisers; split=True adds the two-piece generator backward
    with the decoder-stage range exchanged on its own (on a GPU: on a side stream next to the second piece)"""
    os.environ['FSV2V_EMU'] = '1'
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['FSV_SEG_EARLY_G'] = '0' if serial else '1'        # serial: round 4's order (D exchange in front of the generator pass)
    os.environ['FSV_SEG_SPLIT_ADAM'] = '1' if split_adam else '0'  # round 6: Adam of the decoder range next to the last exchange range
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    torch.set_num_threads(1)
    import model_checks as mc
    from importlib import import_module
    dist.init_process_group('gloo', rank=rank, world_size=world)
    M = mc._model()
    gs = import_module('few-shot-vid2vid_amd.graph_step')
    opt = mc.tiny_opt(ngf=4, ndf=4, nff=4, warp_ref=True, spade_combine=True, fineSize=32, loadSize=32, n_downsample_G=3,
                      n_adaptive_layers=2)
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    model.train()
    opt_G, opt_D = model.build_optimizers(world_size=world, overlap=False, split_backward=split)
    gi = gs.GraphedIteration(model, opt, warmup=1)
    assert gi.segmented and gi.split == bool(split) and gi.pieces == (3 if split == 3 else (2 if split else 1))
    assert gi.seg_early == (not serial) and gi.n_segments() == (3 + (gi.pieces - 1 if split else 0) + (0 if serial else 1) +
                                                               (1 if (split_adam and split is True) else 0))
    tl, ti, rl, ri = mc.synth_pose_inputs(1, 32, 32, 200 + rank, 6)
    data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
    for it in range(2):
        gi(data)
    torch.save(dict(g={n: p.grad.clone() for n, p in model.netG.named_parameters()},
                    p={n: p.detach().clone() for n, p in model.netG.named_parameters()},
                    pD={n: p.detach().clone() for n, p in model.netD.named_parameters()},
                    gD={n: p.grad.clone() for n, p in model.netD.named_parameters() if p.grad is not None},
                    launch=gi.launch_mode(), split_at=opt_G.split_at,
                    total=opt_G.total, split_at2=opt_G.split_at2),
               os.path.join(out_dir, 'split%d%s%s_rank%d.pt' % (int(split), '_serial' if serial else '', '_adam2' if split_adam else '', rank)))
    dist.destroy_process_group()


def test_two_rank_generator_pass_next_to_the_discriminator_exchange_is_the_serial_schedule(emu_lib, tmp_path):
    """round 5: at N > 1 the generator-mode forward pass is a segment of its own between the launch of the discriminator's
    all-reduce and the wait for it (graph_step.GraphedIteration._steps) - two ranks, two iterations with real Adam steps: gradients
    and weights of G and D bit for bit those of round 4's serial order (exchange(D), Adam(D), then the generator pass)"""
    world = 2
    mp.spawn(_split_worker, args=(world, 29661, str(tmp_path), True, True), nprocs=world, join=True)
    mp.spawn(_split_worker, args=(world, 29663, str(tmp_path), True, False), nprocs=world, join=True)
    a0 = torch.load(os.path.join(tmp_path, 'split1_serial_rank0.pt'))
    b0 = torch.load(os.path.join(tmp_path, 'split1_rank0.pt'))
    b1 = torch.load(os.path.join(tmp_path, 'split1_rank1.pt'))
    assert '5 eager segments' in b0['launch'] and '4 eager segments' in a0['launch'], (a0['launch'], b0['launch'])
    for key in ('g', 'p', 'pD', 'gD'):
        for n in a0[key]:
            assert torch.equal(a0[key][n], b0[key][n]), (key, n)
            assert torch.equal(b0[key][n], b1[key][n]), (key, n)                  # replicas in lock-step


def test_two_rank_split_optimiser_step_next_to_the_last_exchange_is_the_one_piece_step(emu_lib, tmp_path):
    """round 6: the generator's Adam + layout refresh as two segments - the decoder-stage range steps while the last range of the
    gradient exchange is in flight, the rest behind it (graph_step.GraphedIteration._steps, FlatAdam.adam_part) - two ranks, two
    iterations with real Adam steps: gradients and weights of G and D bit for bit those of the single Adam segment"""
    world = 2
    mp.spawn(_split_worker, args=(world, 29671, str(tmp_path), True, False, False), nprocs=world, join=True)
    mp.spawn(_split_worker, args=(world, 29673, str(tmp_path), True, False, True), nprocs=world, join=True)
    a0 = torch.load(os.path.join(tmp_path, 'split1_rank0.pt'))
    b0 = torch.load(os.path.join(tmp_path, 'split1_adam2_rank0.pt'))
    b1 = torch.load(os.path.join(tmp_path, 'split1_adam2_rank1.pt'))
    assert '5 eager segments' in a0['launch'] and '6 eager segments' in b0['launch'], (a0['launch'], b0['launch'])
    for key in ('g', 'p', 'pD', 'gD'):
        for n in a0[key]:
            assert torch.equal(a0[key][n], b0[key][n]), (key, n)
            assert torch.equal(b0[key][n], b1[key][n]), (key, n)                  # replicas in lock-step


def test_two_rank_two_piece_backward_equals_whole_backward(emu_lib, tmp_path):
    world = 2
    mp.spawn(_split_worker, args=(world, 29621, str(tmp_path), False), nprocs=world, join=True)
    mp.spawn(_split_worker, args=(world, 29623, str(tmp_path), True), nprocs=world, join=True)
    w0 = torch.load(os.path.join(tmp_path, 'split0_rank0.pt'))
    s0 = torch.load(os.path.join(tmp_path, 'split1_rank0.pt'))
    s1 = torch.load(os.path.join(tmp_path, 'split1_rank1.pt'))
    assert 0 < s0['split_at'] < s0['total'] and w0['split_at'] == 0
    for n in s0['g']:
        assert torch.equal(s0['g'][n], s1['g'][n]) and torch.equal(s0['p'][n], s1['p'][n]), n      # replicas in lock-step
        # same exchanged gradients and same weights after two iterations as with one whole-buffer all-reduce
        scale = max(float(w0['g'][n].abs().max()), 1e-12)
        assert float((s0['g'][n] - w0['g'][n]).abs().max()) <= 1e-6 * scale, n
        assert torch.equal(s0['p'][n], w0['p'][n]), n


def test_two_rank_three_piece_backward_equals_whole_backward(emu_lib, tmp_path):
    """split_backward=3: decoder range | middle range | encoder range exchanged one after the other (on a GPU the first two on a
    side stream next to the following piece) - same exchanged gradients and weights as one whole-buffer all-reduce"""
    world = 2
    mp.spawn(_split_worker, args=(world, 29631, str(tmp_path), False), nprocs=world, join=True)
    mp.spawn(_split_worker, args=(world, 29633, str(tmp_path), 3), nprocs=world, join=True)
    w0 = torch.load(os.path.join(tmp_path, 'split0_rank0.pt'))
    s0 = torch.load(os.path.join(tmp_path, 'split3_rank0.pt'))
    s1 = torch.load(os.path.join(tmp_path, 'split3_rank1.pt'))
    assert 0 < s0['split_at'] < s0['split_at2'] < s0['total'], (s0['split_at'], s0['split_at2'], s0['total'])
    for n in s0['g']:
        assert torch.equal(s0['g'][n], s1['g'][n]) and torch.equal(s0['p'][n], s1['p'][n]), n
        scale = max(float(w0['g'][n].abs().max()), 1e-12)
        assert float((s0['g'][n] - w0['g'][n]).abs().max()) <= 1e-6 * scale, n
        assert torch.equal(s0['p'][n], w0['p'][n]), n


def _amp_worker(rank, world, port, out_dir):
    """--amp O1 across two ranks: rank 1's second gradient overflows; the overflow test runs on the all-reduced buffer, so
    both ranks must skip that step together and come out with identical weights and loss scales"""
    os.environ['FSV2V_EMU'] = '1'
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    torch.set_num_threads(1)
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    flat = import_module('few-shot-vid2vid_amd.flat')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    p = torch.nn.Parameter(torch.randn(300, generator=g))
    o = flat.FlatAdam([p], 1e-2, (0.5, 0.999), world_size=world, overlap=False, loss_scale=(64.0, 2))
    g = torch.Generator().manual_seed(10 + rank)
    log = []
    for it in range(4):
        gr = torch.randn(300, generator=g)
        if it == 1 and rank == 1:
            gr[7] = float('inf')
        o.zero_grad()
        o.flat_g.copy_(gr * float(o.scaler[0]))
        o.step()
        log.append((float(o.scaler[0]), float(o.scaler[1]), float(o.state[0])))
    torch.save(dict(p=o.flat_p.clone(), log=log), os.path.join(out_dir, 'amp%d.pt' % rank))
    dist.destroy_process_group()


def test_two_rank_amp_overflow_is_skipped_on_every_rank(emu_lib, tmp_path):
    world, port = 2, 29617
    mp.spawn(_amp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'amp0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'amp1.pt'))
    assert torch.equal(r0['p'], r1['p'])
    assert r0['log'] == r1['log']
    # (scale, good steps, Adam t): good / overflow -> halve, no step / good / good -> window of 2 reached, double
    assert r0['log'] == [(64.0, 1.0, 1.0), (32.0, 0.0, 1.0), (32.0, 1.0, 2.0), (64.0, 0.0, 3.0)], r0['log']


def _syncbn_worker(rank, world, port, out_dir):
    os.environ['FSV2V_EMU'] = '1'
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    torch.set_num_threads(1)
    import model_checks as mc
    dist.init_process_group('gloo', rank=rank, world_size=world)
    M = mc._model()
    opt = mc.tiny_opt(ngf=4, ndf=4, warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32,
                      n_downsample_G=3, n_adaptive_layers=2)
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    model.train()
    opt_G, opt_D = model.build_optimizers(world_size=world, overlap=False, sync_bn=True)
    opt_G.set_lr(0.0); opt_D.set_lr(0.0)
    tl, ti, rl, ri = [t[rank:rank + 1] for t in mc.synth_pose_inputs(2, 32, 32, 300, opt.input_nc)]
    data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
    d = M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
    gD = opt_D.flat_g.clone()
    g, generated, _ = model(data, save_images=True, mode='generator')
    g = M.loss_backward(opt, g, opt_G, 0)
    bn = {k: v.clone() for k, v in model.netG.state_dict().items() if 'running_' in k}
    torch.save(dict(gD=gD, gG=opt_G.flat_g.clone(), img=generated[0].detach().clone(), bn=bn,
                    g=[float(x.detach()) for x in g if not isinstance(x, int)]), os.path.join(out_dir, 'sbn%d.pt' % rank))
    dist.destroy_process_group()


def _syncbn_single():
    os.environ['FSV2V_EMU'] = '1'
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import model_checks as mc
    M = mc._model()
    opt = mc.tiny_opt(ngf=4, ndf=4, warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32,
                      n_downsample_G=3, n_adaptive_layers=2)
    model = M.create_model(opt)
    mc.fill_state(model.netG); mc.fill_state(model.netD)
    model.train()
    opt_G, opt_D = model.build_optimizers(world_size=1)
    opt_G.set_lr(0.0); opt_D.set_lr(0.0)
    tl, ti, rl, ri = mc.synth_pose_inputs(2, 32, 32, 300, opt.input_nc)
    data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
    M.loss_backward(opt, model(data, mode='discriminator'), opt_D, 1)
    gD = opt_D.flat_g.clone()
    g, generated, _ = model(data, save_images=True, mode='generator')
    M.loss_backward(opt, g, opt_G, 0)
    bn = {k: v.clone() for k, v in model.netG.state_dict().items() if 'running_' in k}
    return dict(gD=gD, gG=opt_G.flat_g.clone(), img=generated[0].detach().clone(), bn=bn)


def test_two_rank_sync_batchnorm_equals_one_process_on_the_whole_batch(emu_lib, tmp_path):
    """opt-in cross-replica BatchNorm (the reference's apex SyncBatchNorm under DDP): two ranks with one sample each produce
    the images, running statistics and (summed) gradients of ONE process that sees both samples"""
    world, port = 2, 29619
    mp.spawn(_syncbn_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'sbn0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'sbn1.pt'))
    one = _syncbn_single()
    img = torch.cat([r0['img'], r1['img']])
    assert float((img - one['img']).abs().max()) <= 2e-5 * float(one['img'].abs().max())
    for k, v in one['bn'].items():
        assert float((r0['bn'][k] - v).abs().max()) <= 1e-5 * max(float(v.abs().max()), 1e-3), k
        assert torch.equal(r0['bn'][k], r1['bn'][k]), k
    # flat_g after the exchange = sum over ranks of the per-rank mean-loss gradients = 2 x the gradient of the batch-mean loss
    for key in ('gD', 'gG'):
        assert torch.equal(r0[key], r1[key])
        ref = 2.0 * one[key]
        rel = float((r0[key] - ref).norm() / ref.norm())
        assert rel <= 2e-3, (key, rel)


def _init_dist_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), FSV2V_EMU='1')
    sys.path.insert(0, ROOT)
    from importlib import import_module
    import fsv2v_amd  # noqa: F401
    integ = import_module('few-shot-vid2vid_amd.integration')
    gpu = integ.init_dist(backend='gloo')
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    torch.save(dict(gpu=gpu, rank=dist.get_rank(), world=dist.get_world_size(), sum=float(t), rnd=torch.rand(3)),
               os.path.join(out_dir, 'init%d.pt' % rank))
    dist.destroy_process_group()


def test_init_dist_replacement(tmp_path):
    """the working stand-in for the reference's util/distributed.py:init_dist (which raises): group joined, per-rank seeds"""
    world, port = 2, 29621
    mp.spawn(_init_dist_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, 'init%d.pt' % i)) for i in range(world)]
    assert [x['rank'] for x in r] == [0, 1] and all(x['world'] == 2 and x['sum'] == 3.0 and x['gpu'] == 0 for x in r)
    assert not torch.equal(r[0]['rnd'], r[1]['rnd'])
