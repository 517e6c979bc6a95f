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
