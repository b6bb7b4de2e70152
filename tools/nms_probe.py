import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch
from ssds.modeling.layers import box
def timeit(fn, n=50, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
B = 64
for N in (256, 1800):
    for valid in (1.0, 0.1, 0.0):
        s = torch.rand(B, N, device="cuda") * (torch.rand(B, N, device="cuda") < valid)
        c = torch.randint(0, 80, (B, N), device="cuda").float()
        xy = torch.rand(B, N, 2, device="cuda") * 400
        wh = torch.rand(B, N, 2, device="cuda") * 60 + 10
        b = torch.cat([xy, xy + wh], -1)
        for nd in (1, 100):
            t = timeit(lambda: box.nms(s, b, c, 0.6, nd, True))
            print("N=%4d valid=%.1f ndet=%3d : %6.1f us" % (N, valid, nd, t))
