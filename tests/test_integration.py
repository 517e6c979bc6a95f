"""Name-binding drop-in for the reference's train.py (only runs where the reference tree is present)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present on this box")


def test_patch_reference_rebinds_the_seams(emu_lib):
    import importlib
    import fsv2v_amd  # noqa: F401
    ref_import.install_shims()
    integ = importlib.import_module('few-shot-vid2vid_amd.integration')
    patched = integ.patch_reference()
    for name in ('models.models.create_model', 'models.loss_collector.loss_backward', 'models.networks.define_G',
                 'models.networks.base_network.resample', 'models.networks.generator.batch_conv'):
        assert name in patched, (name, patched)
    import models.models as mm
    import model_checks as mc
    opt = mc.tiny_opt(warp_ref=True, spade_combine=True, remove_face_labels=True, gpu_ids=[])
    model, flow_net, (og, od) = mm.create_model(opt, 0)
    assert flow_net is None and hasattr(model, 'module')
    assert model.module.lossCollector.loss_names[:2] == ['G_GAN', 'G_GAN_Feat']
    # one reference-style iteration through the re-bound names (emulated kernels, tiny network)
    import models.loss_collector as lc
    tl, ti, rl, ri = mc.synth_pose_inputs(1, 64, 64, 3)
    data = [tl, ti, [None, None], [None, None], rl, ri, None, None, None]
    d = lc.loss_backward(opt, model(data, mode='discriminator'), od, 1)
    g, gen, prev = model(data, save_images=True, mode='generator')
    g = lc.loss_backward(opt, g, og, 0)
    assert len(d) == 4 and len(g) == 10 and len(prev) == 3 and gen[0].shape == (1, 1, 3, 64, 64)
