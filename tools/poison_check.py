"""Debug aid: fill the caching allocator's free blocks with NaN, then run a model check - any kernel that reads memory it
never wrote (and relies on fresh allocations being zero) shows up as NaN / a parity failure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import model_checks as mc
dev = torch.device('cuda:0')
junk = [torch.full((64 << 20,), float('nan'), device=dev) for _ in range(24)]      # 6 GiB of NaN
torch.cuda.synchronize()
del junk
which = sys.argv[1] if len(sys.argv) > 1 else 'face_like'
try:
    if which == 'face_like':
        mc.check_generator(dev, mc.tiny_opt(ngf=8, dataset_mode='fewshot_face', input_nc=1, fineSize=128, loadSize=128), b=2)
    elif which == 'pose':
        mc.check_generator(dev, mc.tiny_opt(ngf=8, nff=8, warp_ref=True, spade_combine=True, fineSize=128, loadSize=128), b=2)
    else:
        mc.check_train_step(dev, mc.tiny_opt(ngf=8, ndf=8, nff=8, warp_ref=True, spade_combine=True, remove_face_labels=True), b=2)
    print('OK', which, {k: v for k, v in os.environ.items() if k.startswith('FSV_')})
except AssertionError as e:
    print('FAIL', which, {k: v for k, v in os.environ.items() if k.startswith('FSV_')}, str(e)[:200])
