"""does the grid build of one half of the batch run under the search of the other?  The config-2 forward
as one call, against the two halves of the batch issued on two streams (each half: build -> stage A -> list kernel; the
second half's build can overlap the first half's search), and with the second stream started one build later.
Usage: python tools/overlap_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_points_amd import synthetic as S  # noqa: E402
from pytorch_points_amd._ext import losses  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, N = 32, 16384
    x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev)
    x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev)
    d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
    i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
    h = B // 2
    halves = [(x1[:h].contiguous(), x2[:h].contiguous(), d1[:h], d2[:h], i1[:h], i2[:h]),
              (x1[h:].contiguous(), x2[h:].contiguous(), d1[h:], d2[h:], i1[h:], i2[h:])]
    s2 = torch.cuda.Stream()

    def whole():
        losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)

    def serial_halves():
        for a in halves:
            losses.nmdistance_forward(*a)

    def two_streams():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        losses.nmdistance_forward(*halves[0])
        with torch.cuda.stream(s2):
            losses.nmdistance_forward(*halves[1])
        cur.wait_stream(s2)

    ref = None
    for name, fn in (("one call, B=32", whole), ("two halves, one stream", serial_halves), ("two halves, two streams", two_streams)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(7):
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        out = (d1.clone(), d2.clone(), i1.clone(), i2.clone())
        if ref is None:
            ref = out
        same = all(torch.equal(a, b) for a, b in zip(ref, out))
        print("%-28s forward %.1f us (median of 7 x 20; min %.1f)  outputs %s" % (name, float(np.median(ts)), min(ts), "identical" if same else "DIFFER"), flush=True)


if __name__ == "__main__":
    main()
