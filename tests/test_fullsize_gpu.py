"""Parity at the configurations that are benchmarked (BASELINE.json configs[0..2]): full network
width (ngf = ndf = nff = 32), full resolution, the batch of the config - the product on the MI355X against the CPU oracle
running on the GPU box's host cores, same seeded inputs and weights.

  C1  fewshot_face 128x128  B=1  adaptive_spade          full D step + G step   (fp32 and fp64 oracle: noise-floor form; four seeds)
  C2  fewshot_face 256x256  B=4  adaptive_spade          generator forward + backward (the config is defined G-only)
  C3  fewshot_pose 512x512  B=2  adaptive_spade + warp_ref + spade_combine   full D step + G step = the bench.py workload
  C4  C3 + --add_face_D (+ VGG19 loss)  B=2 (the per-rank batch of the 8-GPU config)   full D step + G step
  C5  fewshot_street 1024x512  label_nc 35  B=1 (per rank)  adaptive_spade   full D step + G step, in fp32 and in the config's
      stated arithmetic (--amp O1, half-precision kernels: against a whole-iteration oracle run in that arithmetic)

C3, C4 and C5 (fp32) - round 6 - are compared with the UNMODIFIED REFERENCE itself: tests/golden/step_pose_fullsize.pt,
step_pose_face_d_fullsize.pt and step_street_fullsize.pt hold the losses, norms and count sketches (model_checks.sketch) of one full-size reference iteration on the
same seeded inputs, and per quantity the fp32 reference's own distance to the fp64 evaluation (oracle/make_golden.py `fullsize`,
minted in the build container) - no CPU oracle runs on the GPU box for them (model_checks.check_train_step_golden).  C1, C2 and
the `--amp` form of C5 keep the inline oracle runs (C1 needs the element-wise noise floor; the `--amp` arithmetic has no reference).

Tolerances: losses and images 1e-3 relative (BASELINE.json north_star); per-parameter gradients in the relative L2 norm
(model_checks.compare_grads_l2), 1e-2 for the full step.  The oracle runs in fp32 and fp64 (C3: ~11 + ~22 GB of host memory):
a bias gradient is a sum of 10^5 ... 10^6 terms with cancellation, where the fp32 CPU reference itself carries more rounding
error than the kernels' fp64 block sums - the allowance is 4x the fp32 oracle's own distance to the fp64 run.  C3 also checks that the step really went through the tiles and split-K
paths the launch plan is built from - the small-network tests never reach them.  The C1 step is additionally pinned to the
reference itself by tests/golden/step_face_fullwidth.pt (tests/test_golden.py)."""
import pytest
import torch

import model_checks as mc

pytestmark = pytest.mark.gpu

DEV = torch.device('cuda:0')


def _conv():
    import fsv2v_amd  # noqa: F401
    from importlib import import_module
    return import_module('few-shot-vid2vid_amd.conv')


# ---- the configurations, and the oracle runs their tests will ask for (conftest.py hands ORACLE_SPECS to tests/oracle_worker.py,
# which computes the pairs on the host cores while the GPU runs the rest of the suite; a test whose pair is not there computes it
# inline - model_checks.oracle_pair) --------------------------------------------------------------------------------------------
def _c1():
    return mc.make_opt(dataset_mode='fewshot_face', input_nc=1, fineSize=128, loadSize=128, batchSize=1)


def _c3():
    return mc.make_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, fineSize=512, loadSize=512, batchSize=2)


def _c4():
    return mc.make_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, add_face_D=True, no_vgg_loss=False,
                       fineSize=512, loadSize=512, batchSize=2)


def _c5(amp='O0'):
    return mc.make_opt(dataset_mode='fewshot_street', label_nc=35, input_nc=3, aspect_ratio=2.0, fineSize=1024, loadSize=1024,
                       batchSize=1, amp=amp)


ORACLE_SPECS = {
    'test_c1_face_128_full_step': lambda it: [mc.oracle_spec('fp32', _c1(), 1, it.callspec.params['seed'])],
    'test_c1_face_128_full_step_inputs_on_a_kink': lambda it: [mc.oracle_spec('fp32', _c1(), 1, it.callspec.params['seed'])],
    'test_c1_face_128_full_step_fixed_order': lambda it: [mc.oracle_spec('fp32', _c1(), 1, 24)],
    'test_c5_street_1024x512_nc35_amp': lambda it: [mc.oracle_spec('amp', _c5('O1'), 1, 21, loss_scale=1024.0)],
    'test_c5_street_1024x512_nc35_amp_in_the_schedule_bench_py_runs': lambda it: [mc.oracle_spec('amp', _c5('O1'), 1, 21, loss_scale=1024.0)],
}


@pytest.mark.parametrize('seed', [23, 24])
def test_c1_face_128_full_step(hip_lib, seed):
    """C1 in the DEFAULT mode at the 1e-2 gradient bar.  Since round 4 the split-K launches of the fp32 gather-GEMM sum their
    splits in a fixed order (csrc/conv_igemm.hip fsv_split_finish_kernel) - the source of the run-to-run variation round 3 traced
    (tests/c1_kink.py): the step now lands on the same side of every LeakyReLU kink on every run (two runs per seed on hardware:
    1.65e-3 / 1.87e-3 and 2.79e-3 / 2.80e-3 - what still varies is the weight gradients' pixel-split atomics, 1e-4)."""
    opt = _c1()
    worst = mc.check_train_step(DEV, opt, b=1, tol=1e-3, grad_tol=1e-2, seed=seed)
    assert worst < 1e-2, worst


@pytest.mark.parametrize('seed,band', [(21, 5e-2)])
def test_c1_face_128_full_step_inputs_on_a_kink(hip_lib, seed, band):
    """A seed of the same configuration whose input puts ONE pre-activation of a 16-pixel layer (ONE sample, 128x128: the up path of
    the reference-image encoder normalises 16 pixels per channel) within rounding of the LeakyReLU kink: seed 21 - channel 46,
    pixel 6 of ref_img_up_2 is -1e-6 in the oracle and +3e-6 here, 25 ulp from the kink, and its slope (1 or 0.2) carries 3.6 % of
    that layer's weight gradient (profiles/r03_notes.md section 8): 3.58e-2, the same value on every run since the ordered split.
    Both sides are correct fp32 evaluations; no implementation can be held to 1e-2 on these inputs.  Losses and images hold 1e-3
    here too.  Round 5: the kink condition is ASSERTED, not narrated - the product's own pre-activation z = BatchNorm(conv(x)) of
    that layer is captured in the step, and the wide band is admitted only if an element of it lies within 2e-5 of zero (25 ulp
    of the layer's unit-variance values is 3e-6); without such an element the step is held to the plain 1e-2."""
    opt = _c1()
    cap = {'ref_img_up_2.conv': None}
    try:
        worst = mc.check_train_step(DEV, opt, b=1, tol=1e-3, grad_tol=band, seed=seed, capture=cap)
    finally:
        y, net = cap.get('ref_img_up_2.conv'), cap.get('_netG')
    assert y is not None, 'the layer was not captured'
    bn = net.ref_img_up_2.bn
    yy = y.permute(1, 0, 2, 3).reshape(y.shape[1], -1)                                   # [C, N*H*W]: 16 pixels per channel
    assert yy.shape[1] == 16, yy.shape
    z = ((yy - yy.mean(1, keepdim=True)) / torch.sqrt(yy.var(1, unbiased=False, keepdim=True) + 1e-5) *
         bn.weight.detach().double().cpu()[:, None] + bn.bias.detach().double().cpu()[:, None])
    zmin = float(z.abs().min())
    print('seed %d: min |pre-activation| of ref_img_up_2 = %.2e, worst gradient rel L2 %.3e' % (seed, zmin, worst))
    on_kink = zmin < 2e-5
    assert worst < (band if on_kink else 1e-2), (worst, zmin)


def test_c1_face_128_full_step_fixed_order(hip_lib, monkeypatch):
    """C1 at the 1e-2 gradient bar in the fully fixed-order mode (FSV_DETERMINISTIC=1: additionally no reduction split across
    workgroups in the weight gradients, normalisation statistics from their own pass - ten runs of this step from the same state
    are bit-equal, tests/c1_repro.py)."""
    monkeypatch.setenv('FSV_DETERMINISTIC', '1')
    opt = _c1()
    worst = mc.check_train_step(DEV, opt, b=1, tol=1e-3, grad_tol=1e-2, seed=24)
    assert worst < 1e-2, worst


def test_c2_face_256_b4_generator_fwd_bwd(hip_lib):
    opt = mc.make_opt(dataset_mode='fewshot_face', input_nc=1, fineSize=256, loadSize=256, batchSize=4)
    # per-parameter gradients: max-abs 5e-3 + fp32 noise, or - at this size, see model_checks.compare_grads - 2e-2 relative L2
    mc.check_generator(DEV, opt, b=4, tol=1e-3, grads=True, grad_l2_band=2e-2)


def test_c3_pose_512_b2_full_step_is_the_bench_workload(hip_lib):
    conv = _conv()
    opt = _c3()
    conv.start_plan_log()
    try:
        worst = mc.check_train_step_golden(DEV, opt, 'pose_fullsize', tol=1e-3, grad_tol=1e-2)
    finally:
        log = conv.stop_plan_log()
    print('C3 vs the reference fixture: worst generator-gradient sketch distance %.3e of its norm' % worst)
    groups = [e for e in log if e[0] == 'group']                      # ('group', tile, float4 gather, problems)
    log = [e for e in log if e[0] != 'group']
    tiles = {t for t, _, v4 in log if v4}
    assert {0, 1, 2, 4, 9} <= tiles, sorted(tiles)                    # every tile of the plan is exercised by this step
    assert any(s > 1 for _, s, _ in log), "no split-K launch in the 512x512 step"
    assert any(not v4 for _, _, v4 in log), "no scalar-gather launch (3-channel / odd-channel inputs)"
    # grouped launches: the weight-generator bank (16 problems, float4 and - the last layers' data gradients - scalar gather)
    # and the stride-2 data gradients (4 parity classes)
    assert any(n == 16 and v4 for _, _, v4, n in groups) and any(n == 16 and not v4 for _, _, v4, n in groups), groups[:8]
    assert any(n == 4 for _, _, _, n in groups), groups[:8]


def test_c3_pose_512_b2_in_the_schedule_bench_py_runs(hip_lib):
    """The same step issued the way bench.py issues it on one GPU (round 4): the discriminator step on a side stream next to the
    generator-mode forward pass, the G step's real-image pass behind it, the generator's backward in two pieces - against the
    oracle at the same bars.  (The kernels are the ones of the test above - the fused bn_s -> conv_s kernel included, which both
    run; what this adds is the schedule.)"""
    opt = _c3()
    mc.check_train_step_golden(DEV, opt, 'pose_fullsize', tol=1e-3, grad_tol=1e-2, bench_schedule=True)


def test_c3_pose_512_b2_with_the_3x3_spade_fusion(hip_lib, monkeypatch):
    """The bench workload with `FSV_SPADE_CONV3=1` (round 6, opt-in): the level-0 `conv_0` and level-1 `conv_1` of the generator's
    SPADE blocks run as ONE kernel with their modulation + activation (csrc/spade_conv3.hip, the modulated haloed tile in LDS) in
    both generator passes - the whole iteration against the reference's own (fixture) at the bars of the default path, and the
    fused entry point must actually have been the one that ran (both widths, with and without the side output)."""
    from importlib import import_module
    lib = import_module('few-shot-vid2vid_amd.lib')
    monkeypatch.setenv('FSV_SPADE_CONV3', '1')
    seen, real_call = [], lib.call

    def recording_call(name, *a):
        if name == 'fsv_spade_conv3_fwd':
            seen.append((a[24], a[3] is not None))         # (Cout, side output requested)
        return real_call(name, *a)
    monkeypatch.setattr(lib, 'call', recording_call)
    worst = mc.check_train_step_golden(DEV, _c3(), 'pose_fullsize', tol=1e-3, grad_tol=1e-2)
    print('C3 with the 3x3 SPADE fusion vs the reference fixture: worst generator-gradient sketch distance %.3e of its norm' % worst)
    assert {(32, False), (32, True), (64, False), (64, True)} <= set(seen), sorted(set(seen))


def test_c4_pose_512_face_d_vgg(hip_lib):
    """BASELINE.json configs[3] per rank (scripts/pose/train_g8.sh:8-10: the C3 flags + --add_face_D, which brings the VGG19 loss
    with it - loss_collector.py:70-85): full width, 512x512, the per-GPU batch of 2, full D step (netD + netDf) + G step against
    the reference's own iteration (fixture).  VGG19 runs on seeded random weights (no checkpoint in this environment; the reference
    side gets the same tensors through the torchvision stub of oracle/ref_import.py)."""
    opt = _c4()
    mc.check_train_step_golden(DEV, opt, 'pose_face_d_fullsize', tol=1e-3, grad_tol=1e-2)


def test_c5_street_1024x512_nc35_fp32(hip_lib):
    """BASELINE.json configs[4] per rank in fp32 (data/fewshot_street_dataset.py:19-27: W 1024 x H 512, --label_nc 35 one-hot
    labels, --adaptive_spade; one sample per GPU of the 8-GPU batch of 8): full width, full D step + G step against the oracle.
    The config's own arithmetic (--amp O1) is the next test."""
    opt = _c5()
    mc.check_train_step_golden(DEV, opt, 'street_fullsize', tol=1e-3, grad_tol=1e-2)


def test_c5_street_1024x512_nc35_amp(hip_lib):
    """BASELINE.json configs[4] per rank in its stated arithmetic - `--amp O1`, fp16 MFMA (options/base_options.py:127,
    models/models.py:22-26, loss_collector.py:221-224): full width, W 1024 x H 512, 35 one-hot classes, one sample, full D step + G
    step of the product on the half-precision kernels (csrc/conv_h.hip) against a WHOLE-ITERATION run of the oracle in the same
    arithmetic (oracle/np_oracle.amp_conv2d installed into oracle/fsv_oracle.py; same loss scale), summing in fp32 and in fp64.
    Two kinds of bars, both stated in model_checks.check_amp_train_step: (1) EVERY launch that computes in or stores half - the
    gather-GEMMs forward / data / weight gradient, the SPADE kernels on the f16 matrix instructions forward and backward, the fused
    bn_s -> conv_s kernel, the half side outputs of norm-apply / norm-backward / activation-backward, the packed discriminator
    input, both `--amp` Adam steps - is recomputed from its OWN operands with plain torch and held to fp32 summation order (+ one
    half ulp where the kernel rounds at its store): identical inputs, no allowance for the arithmetic's rounding boundaries
    (model_checks.verify_half_launches); (2) the free-running iteration against the oracle as a STATISTICAL bar: image and
    per-parameter gradients in the relative L2 norm within 2 x the fp32 oracle's own distance to its fp64 run (+ the fp32 test's
    1e-3 / 1e-2), losses within 2 x that distance - two correct evaluations of an arithmetic with rounding boundaries are equally
    far from the exact answer in L2, while any max-abs bar is decided by the one activation that crossed a boundary.

    *Parity unpinned against apex*: apex is neither vendored by the reference nor installable here and has no CPU path, so no
    reference output exists for this mode; the oracle states the definition (operands and half-stored activations rounded to
    IEEE half, exact products, fp32 accumulation, fp32 everywhere else) and this test pins the product to it."""
    opt = _c5('O1')
    mc.check_amp_train_step(DEV, opt, b=1, tol=1e-3, grad_tol=1e-2)


def test_c5_street_1024x512_nc35_amp_in_the_schedule_bench_py_runs(hip_lib):
    """configs[4] per rank in its stated arithmetic, issued the way `bench.py --workload street --amp O1` issues it on one GPU
    (discriminator step on a side stream next to the generator-mode forward pass, real-image pass behind it, two-piece backward):
    the same whole-iteration oracle run, the same bars.  *Parity unpinned against apex* (see the test above)."""
    opt = _c5('O1')
    mc.check_amp_train_step(DEV, opt, b=1, tol=1e-3, grad_tol=1e-2, bench_schedule=True)

