"""Localises an error of csrc/ssdk_mbk.hip: runs one 160 -> 960 -> 160 block on a 16x16 map with the projection restricted to
single waves / single chunk pairs (SSDK_MBK_MASKS, debug instance of the kernel) and compares each partial result with the
same partial computed by torch in fp32; then the error structure of the full result by row / column / channel fragment."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]
import torch
import torch.nn.functional as F

from ssds import _native as N
from ssds.modeling.layers import fused_conv as FC
from ssds.modeling.layers.planner import groups_of
from ssds.modeling.nets.mobilenet import InvertedResidual

dtype = torch.bfloat16
torch.manual_seed(5)
cin, cout, h, n, nw = 160, 160, 16, 2, 4
blk = InvertedResidual(cin, cout, 1, 6).eval()
for m in blk.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    if isinstance(m, torch.nn.Conv2d):
        m.weight.data = (m.weight.data * 2).to(dtype).float()
x = torch.randn(n, cin, h, 16).to(dtype)
blk = blk.cuda()
pk = FC.MbPack(groups_of(blk.conv), blk.use_res_connect, dtype)
xd = x.cuda()
xf = xd.float()
# fp32 pieces from the pack (what the kernel computes)
e = F.conv2d(xf, pk.e.w.float().reshape(pk.chid, cin, 1, 1)) + pk.e.bias.view(1, -1, 1, 1)
e = e.clamp(0, 6)
d = (F.conv2d(e, pk.wd.float().permute(2, 0, 1).unsqueeze(1).contiguous(), None, 1, 1, 1, pk.chid) + pk.bd.float().view(1, -1, 1, 1)).clamp(0, 6)
wp = pk.wp.float().reshape(cout, pk.chid)
nch = pk.chid // 16
nchw = (nch + nw - 1) // nw
npair = (nchw + 1) // 2


def expected(hid_sel):
    y = torch.einsum("oh,nhyx->noyx", wp[:, hid_sel], d[:, hid_sel])
    y = y * pk.p.scale.view(1, -1, 1, 1) + pk.p.bias.view(1, -1, 1, 1)
    return y.to(dtype).float() + (xf if pk.residual else 0)


def run(wm, pm):
    os.environ["SSDK_MBK_MASKS"] = "%x,%x" % (wm, pm)
    got = FC.mbconv_native(xd, pk, variant=3)
    torch.cuda.synchronize()
    assert "mbk" in N.last_kernel(), N.last_kernel()
    return got.float()


def relerr(got, want):
    return float((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())


full = run(0xffffffff, 0xffffffff)
want = expected(torch.arange(pk.chid, device="cuda"))
print("full: rel rms err %.4f" % relerr(full, want))
for w in range(nw):
    sel = torch.arange(w * nchw * 16, min((w + 1) * nchw * 16, pk.chid), device="cuda")
    print("wave %d alone: rel rms err %.4f" % (w, relerr(run(1 << w, 0xffffffff), expected(sel))))
for t in range(npair):
    sel = torch.cat([torch.arange((w * nchw + 2 * t) * 16, (w * nchw + min(2 * t + 2, nchw)) * 16, device="cuda") for w in range(nw)])
    print("pair %d alone: rel rms err %.4f" % (t, relerr(run(0xffffffff, 1 << t), expected(sel))))
for w in range(nw):
    for t in (0, 1, npair - 1):
        sel = torch.arange((w * nchw + 2 * t) * 16, (w * nchw + min(2 * t + 2, nchw)) * 16, device="cuda")
        g, wn = run(1 << w, 1 << t), expected(sel)
        print("wave %d pair %d: rel rms err %.4f" % (w, t, relerr(g, wn)))
err = (full - want)
print("error rms by output row :", ["%.3f" % float(err[:, :, y].pow(2).mean().sqrt()) for y in range(h)])
print("error rms by column     :", ["%.3f" % float(err[:, :, :, xx].pow(2).mean().sqrt()) for xx in range(16)])
print("error rms by fragment f :", ["%.3f" % float(err[:, 16 * f:16 * f + 16].pow(2).mean().sqrt()) for f in range(cout // 16)])
print("error rms by image      :", ["%.3f" % float(err[i].pow(2).mean().sqrt()) for i in range(n)])
# one chunk through the whole pipeline: is it the expand, the depthwise or the projection?
os.environ["SSDK_MBK_MASKS"] = "ffffffff,ffffffff"
tiled = FC.mbconv_native(xd, pk, variant=-1).float()
print("tiled kernel vs fp32: rel rms err %.4f" % relerr(tiled, want))
