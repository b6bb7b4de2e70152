#!/usr/bin/env python
"""Soak of every multi-queue mode of the hot path against the in-line path.

    SSDK_LDS_POISON=1 python tools/soak_multiqueue.py --batches 5000

Three HIP queues are live at once, exactly as in bench.py's default mode: the plan's main lane, the plan's side lane
(small heads, ssdk_run_ops_ctx) and the decoder's tail stream (tail_kernel of batch i under the forward of batch
i+1).  Every batch's detections must equal -- bit for bit -- those of the same input computed on ONE stream (side lane
off, no tail stream).  With SSDK_LDS_POISON=1 every kernel additionally starts on NaN-filled LDS, so a read of LDS the
kernel did not write cannot hide behind a friendly previous tenant.  Prints one JSON line; exit code 1 on any mismatch."""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ssds.pytorch_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=5000)
    ap.add_argument("--inputs", type=int, default=12, help="distinct input batches that are cycled through")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    args = ap.parse_args()
    import torch
    from ssds.modeling import nets, ssds
    from ssds.modeling.layers import box
    from ssds.modeling.layers.decoder import Decoder

    torch.manual_seed(8)
    fl = [[5, 7, "Conv:S", "Conv:S", "Conv:S"], [96, 320, 256, 128, 128]]
    nets_outputs, extras, hd = ssds.SSD.add_extras(fl, [6] * 5, 7)
    model = ssds.SSD(nets.MobileNetV2(outputs=nets_outputs), extras, hd, 7).eval().cuda().to(torch.bfloat16)
    for c in model.conf:  # varied scores (the reference init ties everything): the decode stage has real work
        c.weight.data.normal_(0, 0.05)
    anchors = OrderedDict((s, box.generate_anchors(s, [1, 2, 0.5], [2.0, 2.828])) for s in (16, 32, 64, 128, 256))
    xs = [torch.rand(args.batch, 3, args.size, args.size, device="cuda").to(torch.bfloat16) for _ in range(args.inputs)]
    with torch.no_grad():
        plan = model._plan(xs[0])
        assert not isinstance(plan, str), plan
        plan.ctx.set_side_lane(False)  # ---- reference: everything on one stream
        inline = Decoder(0.005, 0.6, 50, 100, True, True)
        want = [tuple(t.clone() for t in inline(*model(x), anchors)) for x in xs]
        torch.cuda.synchronize()
        plan.ctx.set_side_lane(True)   # ---- three queues
        piped = Decoder(0.005, 0.6, 50, 100, True, True).enable_tail_stream()
        bad, done, t0 = 0, 0, time.time()
        while done < args.batches:
            got = [piped(*model(x), anchors) for x in xs]
            piped.wait()
            torch.cuda.synchronize()
            for g, w in zip(got, want):
                bad += 0 if all(torch.equal(a, b) for a, b in zip(g, w)) else 1
            done += len(xs)
    print(json.dumps({"batches": done, "mismatching_batches": bad, "lds_poison": os.environ.get("SSDK_LDS_POISON", "0"),
                      "queues": "main lane + side lane + tail stream", "seconds": round(time.time() - t0, 1)}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
