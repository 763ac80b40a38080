"""Development probe: the stride-2 data gradients of drn_d_38 at the benchmark's batch in the three tilings of the
library option DGRAD_INTERLEAVE (0 = classes one after the other, 1 = interleaved, 2 = row classes with dense stores), each launch timed
alone with HIP events.    python tools/probes/dgrad_s2_ab.py [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
import mcdseg  # noqa: E402
from mcdseg import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
# (Cin, Cout, k, H, W) of the convolution's INPUT: layer2, layer3 conv1 + its 1x1 shortcut, layer4 conv1 + its shortcut
CASES = [(16, 32, 3, 480, 640), (32, 64, 3, 240, 320), (32, 64, 1, 240, 320), (64, 128, 3, 120, 160), (64, 128, 1, 120, 160)]
for cin, cout, k, h, w in CASES:
    g = torch.Generator().manual_seed(1)
    wt = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cout)) ** 0.5).to(dev)
    desc = ops.conv_desc((N, cin, h, w), wt.shape, 2, k // 2, 1)
    packed = ops.PackedWeights()
    wf, wd, _ = packed.get(wt, desc)
    gy = torch.randn(N, cout, desc.Ho, desc.Wo, device=dev)
    gy_cb, gy_bound = ops.split_companion(gy)
    other = torch.randn(N, cin, h, w, device=dev)
    alg = (N * cin * h * w * 4 + N * cout * desc.Ho * desc.Wo * 4) / 1e6  # dx written + the two pieces of dz read
    line = "%3d->%3d %dx%d s2 @ %dx%d  (%.0f MB)" % (cin, cout, k, k, h, w, alg)
    ref = None
    for form in (0, 1, 2):
        mcdseg.set_option("DGRAD_INTERLEAVE", form)
        for addend in (None, other):
            for _ in range(3):
                dx = ops._conv_dgrad(desc, None, wd, dy_cb=gy_cb, dy_bound=gy_bound, w_bound=packed.w_bound, addend=addend)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                dx = ops._conv_dgrad(desc, None, wd, dy_cb=gy_cb, dy_bound=gy_bound, w_bound=packed.w_bound, addend=addend)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            if addend is None:
                if ref is None:
                    ref = dx.clone()
                assert torch.equal(ref.view(torch.int32), dx.view(torch.int32)), "form %d differs" % form
            line += "   form %d%s %.3f ms" % (form, "+add" if addend is not None else "", ms)
    print(line, flush=True)
