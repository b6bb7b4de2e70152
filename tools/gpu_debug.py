import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import cases
from oracle import box_oracle as O
from ssds.modeling.layers import box

np.set_printoptions(linewidth=200, precision=4, suppress=True)
d = cases.decode_inputs("small")
anchors = cases.anchors_for(d["A"], d["stride"], O.generate_anchors)
cls, loc = torch.from_numpy(d["cls"]).cuda(), torch.from_numpy(d["box"]).cuda()
for top_n in (50, 8):
    s, b, c = box.decode(cls, loc, d["stride"], d["thr"], top_n, torch.from_numpy(anchors), False)
    ws, wb, wc = O.decode(d["cls"], d["box"], d["stride"], d["thr"], top_n, anchors, False)
    s = s.cpu().numpy()
    print("top_n", top_n)
    print("got  scores", s[0][:16])
    print("want scores", ws[0][:16])
    flat = d["cls"][0].reshape(-1)
    # map got scores back to indices
    gi = [int(np.nonzero(flat == v)[0][0]) if (flat == v).any() else -1 for v in s[0]]
    wi = [int(np.nonzero(flat == v)[0][0]) for v in ws[0] if v > 0]
    print("got idx ", gi[:24])
    print("want idx", wi[:24])
    print("set equal:", set(gi) == set(wi), "sorted:", bool(np.all(np.diff(s[0]) <= 0)))
    print("npass", int((flat >= np.float32(d["thr"])).sum()))
