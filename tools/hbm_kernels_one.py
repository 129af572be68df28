"""A few launches of every HBM-bound kernel of the MFB train step at the headline shapes (N=512, L=196, O=1000, D=2048,
co-attention hidden 1024), on random operands, for rocprofv3 --pmc passes (tools/profile_r03.sh).  Dropout active (p=0.1)
in the fusion kernels, as in the bench step."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
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
for _ in range(args.reps):
    Y, norm, inv, _ = ops.mfb_fuse_fwd(P, q, N, L, O, seed=123, p_drop=0.1, pbias=pb)          # mfb_fuse_fwd, l2_group_norm, scale_rows
    ops.mfb_fuse_bwd(dY, Y, norm, inv, P, q, N, L, O, seed=123, p_drop=0.1, want_dbias=True, pbias=pb)   # rowdot, coef, mfb_fuse_bwd
    wts, pooled = ops.glimpse_pool_fwd(img, logits, False)
    ops.glimpse_pool_bwd(dpool, img, wts, False, False)
    ops.att_logits_fwd(hid, w2, b2)
    ops.att_logits_bwd(dlog, hid, w2, relu_mask=True)
torch.cuda.synchronize()
