"""Times every HBM-bound kernel of the MFB train step at the headline shapes (N=512, L=196, O=1000, D=2048, hidden 1024) with the
library's hipEvent profiler, for the library VQF_LIB selects:  VQF_LIB=variants/libvqf_X.so python tools/hbm_kernels_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
N, L, O, D, H = 512, 196, 1000, 2048, 1024
g = torch.Generator(device="cuda").manual_seed(5)
rn = lambda *s: torch.randn(s, device="cuda", generator=g)
P, q, pb = rn(N * L, 5 * O), rn(N, 5 * O), rn(5 * O)
dY = rn(N * L, O)
img = torch.relu(rn(N, L, D))
hid = torch.relu(rn(N * L, H))
w2, b2 = rn(2, H) * 0.05, rn(2)
logits = rn(N * L, 2)
dpool = rn(N, 2 * D)
dlog = rn(N * L, 2)


def once():
    Y, norm, inv, _ = ops.mfb_fuse_fwd(P, q, N, L, O, seed=123, p_drop=0.1, pbias=pb, normalise=False)
    ops.mfb_fuse_bwd(dY, Y, norm, inv, P, q, N, L, O, seed=123, p_drop=0.1, want_dbias=True, pbias=pb)
    wts, pooled = ops.glimpse_pool_fwd(img, logits, False)
    ops.glimpse_pool_bwd(dpool, img, wts, False, False)
    ops.att_logits_fwd(hid, w2, b2)
    ops.att_logits_bwd(dlog, hid, w2, relu_mask=True)


for _ in range(2):
    once()
torch.cuda.synchronize()
ops.prof_reset(); ops.prof_enable(True)
R = 5
for _ in range(R):
    once()
torch.cuda.synchronize()
ops.prof_enable(False)
alg = {"mfb_fuse_fwd": 4.0 * N * L * 5 * O + 4.0 * N * L * O, "mfb_fuse_bwd": 8.0 * N * L * 5 * O + 8.0 * N * L * O,
       "glimpse_pool_fwd": 4.0 * N * L * D, "glimpse_pool_bwd": 4.0 * N * L * D, "att_logits_fwd": 4.0 * N * L * H,
       "att_logits_bwd": 8.0 * N * L * H}
print("library:", os.environ.get("VQF_LIB", "default"))
for k, (n, ms) in sorted(ops.prof_report().items()):
    if k in alg:
        print("%-18s %7.4f ms  %6.0f GB/s" % (k, ms / n, alg[k] / (ms / n) / 1e6))
