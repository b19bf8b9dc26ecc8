"""debug (ON the GPU box): the two-pass records after a hook call -- header, per-wave heads, per-plane cost sums"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import mc_cnn_amd as mc
from oracle import cpu_oracle as oracle
from util import smooth_pair, raw_volumes
H, W, D = 90, 300, 9
x0, x1 = smooth_pair(H, W, 8, seed=H)
x0c, x1c = oracle.cross(x0, 14, 0.02), oracle.cross(x1, 14, 0.02)
vl, vr = raw_volumes(D, H, W, seed=13)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
out = torch.full((1, D, H, W), -7.0, device="cuda")
mc.adcensus.cbca_cfg(dev(x0c), dev(x1c), dev(vl), out, -1, form=10)
torch.cuda.synchronize()
lib = mc._lib.lib
cs = lib.mc_cbca_scratch_bytes(H, W); need = cs + lib.mc_cbca_plan_bytes(D, H, W) + 512 + 4 * D * H * W
off = (cs + 255) // 256 * 256
sc = mc.adcensus._scratch_for(out.device, need)
words = sc[off:off + lib.mc_cbca_plan_bytes(D, H, W) // 4 * 4].view(torch.int32).cpu().numpy()
print("header", words[:8])
gx, gy = -(-W // 252), -(-H // 8)
recs = words[64:64 + gx * gy * D * 1024].reshape(D, gy * gx, 1024)
cnt = recs[:, :, 0]; cost = recs[:, :, 1].copy().view(np.float32)
print("counts plane 0", cnt[0]); print("cost plane 0", cost[0]); print("plane sums", cost.sum(1), "limit", 2.0 * H * W, "max count", cnt.max())
