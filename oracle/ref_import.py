"""Import the REAL reference (NVlabs/few-shot-vid2vid at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  This module exists to (a) pin the oracle restatement in
``oracle/fsv_oracle.py`` against the reference's own Python and (b) mint the golden
fixtures under ``tests/golden/`` (see ``oracle/make_golden.py``).  /root/reference
only exists in the build container, so nothing that runs on the GPU box may import
this file.

The reference targets PyTorch 1.2 + apex + torchvision; the shims below are the
minimal environment repairs listed in SURVEY.md section 8(c):

  * apex.parallel.SyncBatchNorm -> torch.nn.BatchNorm2d (identical maths in one
    process; reference call site models/networks/normalization.py:15)
  * torchvision / cv2 / dominate stubs (models/networks/vgg.py:10, util/util.py:12,
    util/html.py:8)
  * fractions.gcd (models/trainer.py:14)
  * Adam betas int -> float (models/base_model.py:45-48)
  * Tensor.cuda / Module.cuda / get_device / torch.cuda.*Tensor on a CPU-only build
    (models/networks/base_network.py:26,30,32; models/input_process.py:40,74)
"""
import argparse
import fractions
import math
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("FSV_REFERENCE_ROOT", "/root/reference")
_installed = False


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models", "networks"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_shims():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    # --- apex -------------------------------------------------------------------
    apex = _stub("apex")
    apex.parallel = _stub("apex.parallel", SyncBatchNorm=torch.nn.BatchNorm2d)
    # --- torchvision (random-weight VGG19 so add_face_D / VGG loss can run) ---------
    def vgg19(pretrained=False):
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M',
               512, 512, 512, 512, 'M']
        layers, cin = [], 3
        g = torch.Generator().manual_seed(19)       # same stream as few-shot-vid2vid_amd/vgg.py:random_vgg19_weights
        for v in cfg:
            if v == 'M':
                layers.append(torch.nn.MaxPool2d(2, 2))
            else:
                conv = torch.nn.Conv2d(cin, v, 3, padding=1)
                with torch.no_grad():
                    conv.weight.copy_(torch.randn((v, cin, 3, 3), generator=g) * math.sqrt(2.0 / (9 * cin)))
                    conv.bias.zero_()
                layers += [conv, torch.nn.ReLU(inplace=False)]
                cin = v
        net = types.SimpleNamespace()
        net.features = torch.nn.Sequential(*layers)
        return net
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models", vgg19=vgg19)
    tv.transforms = _stub("torchvision.transforms")
    # --- misc absent packages -------------------------------------------------------
    _stub("cv2")
    dom = _stub("dominate")
    dom.tags = _stub("dominate.tags")
    if not hasattr(fractions, "gcd"):
        fractions.gcd = math.gcd
    # --- Adam with int beta ---------------------------------------------------------
    _Adam = torch.optim.Adam
    if not getattr(_Adam, "_fsv_patched", False):
        class Adam(_Adam):
            _fsv_patched = True

            def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), **kw):
                super().__init__(params, lr=lr, betas=(float(betas[0]), float(betas[1])), **kw)
        torch.optim.Adam = Adam
    # --- CPU-only build: make .cuda() an identity -------------------------------------
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        torch.Tensor.get_device = lambda self: 0
        torch.cuda.set_device = lambda *a, **k: None
        torch.cuda.FloatTensor = torch.FloatTensor
        torch.cuda.ByteTensor = torch.ByteTensor
        torch.cuda.manual_seed = lambda *a, **k: None
        torch.cuda.manual_seed_all = lambda *a, **k: None
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _installed = True


def install_flownet_shims(use_ref_kernels=True):
    """The reference's FlowNet2 python (models/networks/flownet2_pytorch) imports three compiled CUDA extensions.  They
    cannot be built with nvcc here, so the *compiled modules* - and only those - are replaced by python modules backed by
    the reference's OWN kernel templates compiled for the host from the reference tree (oracle/build_ref.py ->
    oracle/flownet_ref.py); the reference's autograd Function / Module wrappers and all network code then run unmodified on
    CPU tensors.  use_ref_kernels=False (or no reference tree / prebuilt library): the Python restatements of the kernels in
    oracle/flownet_oracle.py instead."""
    install_shims()
    from oracle import flownet_oracle as FO
    K = FO
    if use_ref_kernels:
        from oracle import flownet_ref
        if flownet_ref.load() is not None:
            K = flownet_ref

    def resample_fwd(input1, input2, output, kernel_size):
        assert kernel_size == 1
        output.copy_(K.resample2d(input1, input2))

    def channelnorm_fwd(input1, output, norm_deg):
        assert norm_deg == 2
        output.copy_(K.channelnorm(input1))

    def correlation_fwd(input1, input2, rbot1, rbot2, output, pad_size, kernel_size, max_displacement, stride1, stride2,
                        corr_multiply):
        res = K.correlation(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2)
        output.resize_(res.shape).copy_(res)
    _stub("resample2d_cuda", forward=resample_fwd)
    _stub("channelnorm_cuda", forward=channelnorm_fwd)
    _stub("correlation_cuda", forward=correlation_fwd)
    return K.__name__


def build_flownet2():
    install_flownet_shims()
    from models.networks.flownet2_pytorch import models as flownet2_models
    net = flownet2_models.FlowNet2()
    net.eval()
    return net


def make_opt(argv):
    """Build the reference's ``opt`` Namespace from a flag list, without touching disk.

    Mirrors options/base_options.py:134-225 (gather_options + parse) minus save_options.
    """
    install_shims()
    from options.train_options import TrainOptions
    import importlib
    to = TrainOptions()
    parser = argparse.ArgumentParser()
    parser = to.initialize(parser)
    known, _ = parser.parse_known_args(argv)
    mode = known.dataset_mode
    defaults = {
        'fewshot_pose': dict(label_nc=0, input_nc=6, aspect_ratio=0.5),
        'fewshot_face': dict(label_nc=0, input_nc=1, aspect_ratio=1.0),
        'fewshot_street': dict(label_nc=20, input_nc=3, aspect_ratio=2.0),
    }[mode]
    # dataset flag injection (data/fewshot_*_dataset.py modify_commandline_options); the dataset
    # modules import PIL/cv2 machinery, so the flags are restated here instead of imported.
    parser.add_argument('--label_nc', type=int, default=defaults['label_nc'])
    parser.add_argument('--input_nc', type=int, default=defaults['input_nc'])
    parser.add_argument('--aspect_ratio', type=float, default=defaults['aspect_ratio'])
    if mode == 'fewshot_pose':
        parser.add_argument('--pose_type', type=str, default='both')
        parser.add_argument('--remove_face_labels', action='store_true')
        parser.add_argument('--refine_face', action='store_true')
        parser.add_argument('--basic_point_only', action='store_true')
    opt = parser.parse_args(argv)
    opt.isTrain = True
    ids = [int(s) for s in str(opt.gpu_ids).split(',')]
    opt.gpu_ids = [i for i in ids if i >= 0]
    return opt


def build_model(argv, temporal=False):
    """create the reference Vid2VidModel on CPU (models/models.py:16-38 without WrapModel)."""
    opt = make_opt(argv)
    from util.visualizer import Visualizer
    Visualizer.vis_print = staticmethod(lambda opt, message: None)   # no disk writes (util/visualizer.py:207-212)
    from models.vid2vid_model import Vid2VidModel
    model = Vid2VidModel()
    model.initialize(opt, 0)
    if temporal:
        model.init_temporal_model()
    return opt, model
