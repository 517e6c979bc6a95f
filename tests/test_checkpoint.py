"""Checkpoint compatibility with the reference (only runs where the reference tree is present): files written by one side
load into the other with `load_state_dict(strict=True)` semantics - same file names, keys and shapes
(models/base_model.py:51-93, 219-243)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present on this box")

FLAGS = ('--dataset_mode fewshot_pose --aspect_ratio 1 --fineSize 64 --loadSize 64 --adaptive_spade --warp_ref --spade_combine '
         '--remove_face_labels --no_flow_gt --add_face_D --lambda_temp 1 --gpu_ids -1 --ngf 8 --ndf 8 --nff 8 --batchSize 1')


def test_checkpoints_round_trip_with_the_reference(tmp_path):
    import model_checks as mc
    M = mc._model()
    opt_ref, ref = ref_import.build_model(FLAGS.split() + ['--checkpoints_dir', str(tmp_path), '--name', 'rt'], temporal=True)
    opt = mc.make_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True, remove_face_labels=True, add_face_D=True,
                      lambda_temp=1.0, no_vgg_loss=False, batchSize=1)
    opt.checkpoints_dir, opt.name = str(tmp_path), 'rt'
    mine = M.create_model(opt)
    mine.init_temporal_model()
    nets = (('netG', 'G'), ('netD', 'D'), ('netDf', 'Df'), ('netDT', 'DT'))
    # product -> reference
    for attr, _ in nets:
        mc.fill_state(getattr(mine, attr), scale=0.8)
    mine.save_networks('latest')
    for attr, label in nets:
        path = os.path.join(str(tmp_path), 'rt', 'latest_net_%s.pth' % label)
        assert os.path.isfile(path), path
        getattr(ref, attr).load_state_dict(torch.load(path))           # strict: identical keys and shapes
        a, b = getattr(ref, attr).state_dict(), getattr(mine, attr).state_dict()
        assert all(torch.equal(a[k], b[k]) for k in b)
    # reference -> product
    for attr, label in nets:
        mc.fill_state(getattr(ref, attr), scale=1.1)
        ref.save_network(getattr(ref, attr), label, 7, [])
        assert mine.load_network(getattr(mine, attr), label, 7) == set()
        a, b = getattr(ref, attr).state_dict(), getattr(mine, attr).state_dict()
        assert all(torch.equal(a[k], b[k]) for k in a)
    # a checkpoint written before init_temporal_model: the temporal layers stay uninitialised and are reported
    opt2 = mc.make_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True, remove_face_labels=True)
    opt2.checkpoints_dir, opt2.name = str(tmp_path), 'rt'
    early = M.create_model(opt2)
    early.save_network(early.netG, 'G', 'early')
    missing = mine.load_network(mine.netG, 'G', 'early')
    assert any('img_prev_embedding' in m for m in missing) and mine.load_network(mine.netG, 'G', 'nope') is None
