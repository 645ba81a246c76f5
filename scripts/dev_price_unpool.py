"""Dev pricing (DEV library): what removing the un-pool pass could buy at most.  The fine-tune step (16 clips x T = 16 = 256
windows) with and without the three unpool8_rows launches; without them the consumers (wgrad_patch's dY slabs, the dense /
masked input-gradient kernels) read the gradient images an earlier step left behind -- same bytes from HBM, real-valued
data -- so the difference is the un-pool kernels' own time; RGP_CP_ABLATE=1 additionally feeds the patch kernels' plane
fetches from one L2-resident slab (no HBM reads of their input images: forward AND input-gradient kernels).
usage: python scripts/dev_with_lib.py recurrent_gaze_prediction_amd/librgp_hip_dev.so scripts/dev_price_unpool.py"""
import os, time
import torch
from recurrent_gaze_prediction_amd.finetune import EndToEndGaze
dev = torch.device('cuda:0')
B, T = 16, 16
m = EndToEndGaze(B, T, dtype='bf16', device=dev, max_windows=B * T, seed=1)
g = torch.Generator(device=dev); g.manual_seed(5)
video = torch.rand(B * T, 16, 112, 112, 3, device=dev, generator=g) - 0.5
gt = torch.rand(B, T, 49, 49, device=dev, generator=g) + 1e-3
gt = (gt / gt.sum((-1, -2), keepdim=True)).contiguous()
def timed(k=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): m.train_step(video, gt, 1e-4)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
for _ in range(3): m.train_step(video, gt, 1e-4)
for rnd in range(3):
    for label, env in (('as built', {}), ('no un-pool launches', {'RGP_UNPOOL_SKIP': '1'}),
                       ('no un-pool launches, patch kernels fetch planes from L2', {'RGP_UNPOOL_SKIP': '1', 'RGP_CP_ABLATE': '1'})):
        for k in ('RGP_UNPOOL_SKIP', 'RGP_CP_ABLATE'): os.environ.pop(k, None)
        os.environ.update(env)
        print('%-60s %.3f ms per step' % (label, timed()), flush=True)
for k in ('RGP_UNPOOL_SKIP', 'RGP_CP_ABLATE'): os.environ.pop(k, None)
