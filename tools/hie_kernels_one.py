"""A few launches of the four streaming passes and the two affinity products of HieCoAtten's ladder (csrc/hie.hip) at BASELINE config 4's shapes (N = 256,
L = 196, E = 512, T = 14), dropout active (p = 0.5) in the two tanh passes, for rocprofv3 --pmc passes (tools/profile_r04.sh hie)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
ops = vqa_amd.ops
vqa_amd.lib.load()
N, L, E, T = 256, 196, 512, 14
M, MT = N * L, N * T
g = torch.Generator(device="cuda").manual_seed(7)
rn = lambda *s: torch.randn(s, device="cuda", generator=g)
CI, CQ, dCI = rn(M, 2 * E), rn(MT, 2 * E), torch.empty((M, 2 * E), device="cuda")
C3 = torch.tanh(rn(N, T, L))
S = ops.hie_chunks(N, L)
part, wpart = torch.empty((S, MT, E), device="cuda"), torch.empty((S * N, E + 4), device="cuda")
Hv, dl, whv, dti = torch.empty((M, E), device="cuda"), rn(M), rn(E), rn(MT, E)
cpart = torch.empty((S * N, 3 * E + 4), device="cuda")      # round 5: the partial rows of the three backward passes in one buffer
wpart = cpart[:, 2 * E:]
for _ in range(args.reps):
    Cf = ops.hie_affinity(CQ[:, :E], CI[:, :E], N, L, T, epi=1, drop=(None, 77, 0.5))          # round 5: C = dropout(tanh(Cq Cv^T))
    ops.hie_hv_fwd(CI[:, E:], C3, CQ[:, E:], (None, 123, 0.5), N, L, T, Hv, part)
    ops.hie_head_bwd(Hv, dl, whv, C3, (None, 123, 0.5), N, L, T, dCI[:, E:], part, wpart)
    ops.hie_affinity(dti, CI[:, E:], N, L, T, x2=CQ[:, E:], y2=dCI[:, E:], epi=2, yprev=Cf, drop=(None, 77, 0.5))   # dC, two pairs
    ops.hie_rank_add(dCI[:, E:], C3, dti, N, L, T, dCI[:, E:], colpart=cpart[:, E:2 * E])
    ops.hie_rank_left(C3, CQ[:, :E], CI[:, :E], N, L, T, dCI[:, :E], part, colpart=cpart[:, :E])
torch.cuda.synchronize()
