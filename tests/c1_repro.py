"""Run-to-run reproducibility of the C1 full-size step (fewshot_face 128x128, B=1, full width) and its distance to the oracle.

The product's D step + G step is run N times from the SAME state (fresh model, same seeded weights, u / v, inputs); reported per
run: the worst per-parameter relative-L2 distance of the generator's gradients to the fp64 oracle (the quantity
tests/test_fullsize_gpu.py::test_c1 bounds) and which parameter it belongs to, plus whether the gradients of the run are bit-equal
to those of run 0.  FSV_DETERMINISTIC=1 switches every split reduction off (fixed-order sums).

    python tests/c1_repro.py [N [seed]]          (on the GPU box; lives under tests/ because it uses the oracle as its checker)
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import model_checks as mc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
M = mc._model()
O = mc.O
opt = mc.make_opt(dataset_mode='fewshot_face', input_nc=1, fineSize=128, loadSize=128, batchSize=1)
b, seed = 1, (int(sys.argv[2]) if len(sys.argv) > 2 else 21)
h, w = int(opt.fineSize / opt.aspect_ratio), opt.fineSize
nl = opt.label_nc if opt.label_nc != 0 else opt.input_nc
data = mc.synth_pose_inputs(b, h, w, seed, nl)
data = mc.with_n_shot(data, opt.n_shot, b, h, w, seed, nl)
cfg = O.cfg_from_opt(opt)
ref = None
first = None
rows = []
for run in range(N):
    torch.manual_seed(0)
    model = M.create_model(opt)
    sdG0, sdD0 = mc.fill_state(model.netG), mc.fill_state(model.netD)
    if ref is None:
        r64 = mc._oracle_iteration(sdG0, sdD0, cfg, data, torch.float64, None, None, [None, None], [None, None], None)
        ref = {k: v.detach().double() for k, v in r64[3].items() if v is not None}        # name -> gradient (fp64 oracle)
        assert ref, 'the oracle returned no generator gradients'
        mags = [float(g.norm()) for g in ref.values()]
        floor = 1e-2 * float(np.median(mags))
    model = model.to(dev).train()
    opt_G, opt_D = model.build_optimizers()
    opt_G.set_lr(0.0); opt_D.set_lr(0.0)
    tl, ti, rl, ri = [t.to(dev) for t in data]
    data_list = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
    M.loss_backward(opt, model(data_list, mode='discriminator'), opt_D, 1)
    g_losses, generated, prev = model(data_list, save_images=True, mode='generator')
    M.loss_backward(opt, g_losses, opt_G, 0)
    grads = {n: p.grad.detach().double().cpu() for n, p in model.netG.named_parameters() if p.grad is not None and n in ref}
    assert len(grads) > 100, (len(grads), list(ref)[:3])
    worst, wname = 0.0, ''
    for n, g in grads.items():
        e = float((g - ref[n]).norm()) / max(float(ref[n].norm()), floor, 1e-12)
        if e > worst:
            worst, wname = e, n
    if first is None:
        first = grads
    same = all(torch.equal(grads[n], first[n]) for n in grads)
    ndiff = sum(0 if torch.equal(grads[n], first[n]) else 1 for n in grads)
    rows.append({'run': run, 'worst_rel_l2': round(worst, 5), 'param': wname, 'bit_equal_to_run0': same, 'params_differing': ndiff})
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({'deterministic_env': os.environ.get('FSV_DETERMINISTIC', '0'), 'runs': N, 'seed': seed,
                  'distinct_worst': sorted({r['worst_rel_l2'] for r in rows}),
                  'all_bit_equal': all(r['bit_equal_to_run0'] for r in rows)}))
