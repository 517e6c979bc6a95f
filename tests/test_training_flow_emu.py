"""Drop-in training-loop flow of the reference around the hot path (emulated kernels, tiny networks):

* train.py:36 keeps the optimiser handles it got from create_model for the whole run while models.update_models calls
  model.module.init_temporal_model() at epoch niter_single + 1 (models/models.py:61-72): the SAME handles must keep stepping
  live parameters afterwards, including the branches the temporal model adds;
* --continue_train / a start epoch past niter_single through integration.create_model: checkpoints are loaded
  (vid2vid_model.py:44, base_model.py:229-243), the model starts temporal (base_model.py:213-215), and a pre-temporal
  checkpoint initialises flow_network_temp from flow_network_ref (base_model.py:234-235)."""
import argparse
import importlib

import torch

import model_checks as mc

CPU = torch.device('cpu')


def _opt(tmp_path, **kw):
    base = dict(ngf=4, ndf=4, nff=4, warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=32, loadSize=32,
                n_downsample_G=3, n_adaptive_layers=2, gpu_ids=[], lambda_temp=1.0)
    base.update(kw)
    opt = mc.tiny_opt(**base)
    return argparse.Namespace(**{**vars(opt), 'checkpoints_dir': str(tmp_path), 'name': 'run', 'which_epoch': 'latest',
                                 'continue_train': False, 'load_pretrain': ''})


def _frame(seed, prev=None):
    tl, ti, rl, ri = mc.synth_pose_inputs(1, 32, 32, seed, 6)
    p = prev if prev is not None else [None, None, None]
    return [tl, ti, [None, None], [None, None], rl, ri, p[0], p[1], p[2]]


def _iteration(M, opt, model, og, od, data):
    d = M.loss_backward(opt, model(data, mode='discriminator'), od, 1)
    g, _, prev = model(data, mode='generator')
    M.loss_backward(opt, g, og, 0)
    return prev


def test_create_model_handles_survive_init_temporal_model(emu_lib, tmp_path):
    import fsv2v_amd  # noqa: F401
    integ = importlib.import_module('few-shot-vid2vid_amd.integration')
    M = mc._model()
    opt = _opt(tmp_path)
    model, _, (og, od) = integ.create_model(opt, 0)
    m = model.module
    prev = _iteration(M, opt, model, og, od, _frame(1))
    m.init_temporal_model()                                     # what models.update_models does at epoch niter_single + 1
    assert m.optimizer_G is og and m.optimizer_D is od           # identity kept: train.py's handles are the live optimisers
    assert og.generation == 1 and float(og.state[0]) == 0.0     # fresh Adam state, like the reference's new optimisers
    assert any(p is q for p in m.netDT.parameters() for q in od.params)
    before = {n: p.detach().clone() for n, p in m.netG.named_parameters()}
    before_dt = [p.detach().clone() for p in m.netDT.parameters()]
    _iteration(M, opt, model, og, od, _frame(2, prev))          # second frame: previous-frame branch + temporal discriminator
    moved = {n: float((p.detach() - before[n]).abs().max()) for n, p in m.netG.named_parameters()}
    assert moved['conv_img.weight'] > 0 and moved['up_0.conv_0.weight_orig'] > 0
    new_branch = [n for n in moved if n.startswith('img_prev_embedding')]
    assert new_branch and all(moved[n] > 0 for n in new_branch if n.endswith('weight')), new_branch[:3]
    assert any(float((p.detach() - b).abs().max()) > 0 for p, b in zip(m.netDT.parameters(), before_dt))


def test_continue_train_round_trip_and_late_start(emu_lib, tmp_path):
    import fsv2v_amd  # noqa: F401
    integ = importlib.import_module('few-shot-vid2vid_amd.integration')
    M = mc._model()
    opt = _opt(tmp_path)
    model, _, (og, od) = integ.create_model(opt, 0)
    _iteration(M, opt, model, og, od, _frame(1))
    model.module.save_networks('latest')                        # a pre-temporal checkpoint
    want = {k: v.detach().clone() for k, v in model.module.netG.state_dict().items()}
    # resume in the single-frame phase: identical weights, flat buffers and cached layouts follow them
    opt2 = argparse.Namespace(**{**vars(opt), 'continue_train': True})
    model2, _, (og2, od2) = integ.create_model(opt2, 1)
    m2 = model2.module
    assert not m2.temporal
    for k, v in m2.netG.state_dict().items():
        assert torch.equal(v, want[k]), k
    assert torch.equal(m2.netG.conv_img.weight.detach().reshape(-1),
                       og2.flat_p[[o for (o, n), p in zip(og2.offsets, og2.params) if p is m2.netG.conv_img.weight][0]:][:m2.netG.conv_img.weight.numel()])
    # resume past niter_single: temporal from the start, netDT in the D optimiser, flow_network_temp taken from flow_network_ref
    opt3 = argparse.Namespace(**{**vars(opt), 'continue_train': True, 'sep_flow_prev': True})
    model3, _, (og3, od3) = integ.create_model(opt3, opt.niter_single + 1)
    m3 = model3.module
    assert m3.temporal and m3.netDT is not None
    assert any(p is q for p in m3.netDT.parameters() for q in od3.params)
    ref_sd, tmp_sd = m3.netG.flow_network_ref.state_dict(), m3.netG.flow_network_temp.state_dict()
    same = [k for k in ref_sd if k in tmp_sd and ref_sd[k].shape == tmp_sd[k].shape]
    assert same and all(torch.equal(ref_sd[k], tmp_sd[k]) for k in same)
    for k in ('conv_img.weight', 'up_0.conv_0.weight_orig'):
        assert torch.equal(m3.netG.state_dict()[k], want[k]), k
    prev = _iteration(M, opt3, model3, og3, od3, _frame(3))
    _iteration(M, opt3, model3, og3, od3, _frame(4, prev))      # and it trains


def test_patched_batch_conv_honours_stride(emu_lib):
    import torch.nn.functional as F
    import fsv2v_amd  # noqa: F401
    integ = importlib.import_module('few-shot-vid2vid_amd.integration')
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 10, 12, generator=g)
    w = torch.randn(2, 12, 8, 3, 3, generator=g) * 0.2
    b = torch.randn(2, 12, generator=g)
    for stride in (1, 2):
        ref = torch.cat([F.conv2d(x[i:i + 1], w[i], b[i], padding=1, stride=stride) for i in range(2)])
        got = integ.batch_conv(x, w, b, stride=stride)
        assert float((got - ref).abs().max()) < 1e-5
    import pytest
    with pytest.raises(NotImplementedError):
        integ.batch_conv(x, w, b, stride=0.5)
    with pytest.raises(NotImplementedError):
        integ.batch_conv(x, w, b, group_size=4)
