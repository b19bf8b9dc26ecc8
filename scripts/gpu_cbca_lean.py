"""GPU: cross-based aggregation of a textured pair at 1000x1500x256 (the specified north-star regime) -- the strip kernel against the
classify / lean / list kernels (hook forms 1, 8, 9), rows per wave and prefetch depth varied; run under rocprofv3 --kernel-trace, the
per-kernel times come from the run's rocprofv3 kernel trace (scripts/rocpd_summary.py).  Also prints host-timed ms per call (incl. cbca_pack)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
mc = importlib.import_module("mc-cnn_amd")
from util import smooth_pair, natural_pair
from bench import same_bits_dev
A = mc.adcensus
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
H, W, D, L1, tau1 = 1000, 1500, 256, 14, 0.02
if "--kitti" in sys.argv:
    H, W, D, L1, tau1 = 370, 1226, 228, 14, 0.02
mk = natural_pair if "--natural" in sys.argv else smooth_pair
x0, x1 = mk(H, W, D, seed=1234)
xb = dev(np.stack([x0, x1]))[:, None]
x0c = torch.empty((1, 4, H, W), device="cuda"); x1c = torch.empty_like(x0c)
A.cross(xb[0:1], x0c, L1, tau1); A.cross(xb[1:2], x1c, L1, tau1)
vin = torch.rand((1, D, H, W), device="cuda")
ref = torch.empty_like(vin)
A.cbca_cfg(x0c, x1c, vin, ref, -1, form=1)   # (the strip kernel: pinned against the reference's kernels by tests/test_gpu_fullsize.py)
reps = 4
def run(tag, **kw):
    o = torch.full_like(vin, -7.0)
    fn = lambda: A.cbca_cfg(x0c, x1c, vin, o, -1, **kw)
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print("%-44s %.3f ms/call (incl. pack)  same bits %s" % (tag, (time.time() - t0) * 1000 / reps, same_bits_dev(o, ref)), flush=True)
run("strip kernel", form=1)
# (a pass that reads the list must use the rows per wave the list was written for: every variant lists first -- form 8 -- the kernels'
# own times are in the rocprofv3 trace, per kernel name and grid size)
# every variant lists first (form 8): a pass that reads the list must use the rows per wave the list was written for
# lean_variant: bit 2 the listed outputs in a launch of their own, bit 4 one band of rows per XCD, bits 5 / 6 non-temporal loads / stores
for rb in (2, 4, 8):
    for variant, tag in ((0, "address order"), (16, "band per XCD"), (64, "address order, nt stores"), (4, "address order, own list launch")):
        run("classify + lean<R=%d, %s>" % (rb, tag), form=8, rb=rb, d0=variant)
