"""Where does the linked fp32 co-attention head's input gradient (dYs) differ from fp64?  (round 5 node checks: 1.8e-4 against a
torch-fp32 noise of 4.5e-7 at config 3's shapes.)  Replays the head's backward piece by piece on recorded operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import recipe
from cases import make_cfg
import vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
F = vqa_amd.functions

case = dict(name="c3g", salt=83, N=512, model_name="mhb_coAtt", glove=False, H=1024, E=300, D=2048, L=196, V=1000, A=1000, T=14)
cfg = make_cfg(case)
model = vqa_amd.MHBCoAtt(cfg)
model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), case["salt"])) for k, v in model.state_dict().items()})
model = model.cuda().train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
img = torch.relu(torch.randn((512, 196, 2048), generator=torch.Generator().manual_seed(1234))).cuda()
q = torch.randint(1, 1000, (512, 14), generator=torch.Generator().manual_seed(1235)).cuda()
soft = torch.softmax(torch.randn((512, 1000), generator=torch.Generator().manual_seed(1236)), 1).cuda()

rec = {}
orig = F.AttHeadFn.apply
def wrap(*args):
    out = orig(*args)
    if len(args) > 10 and args[10] is not None:
        rec["args"] = args
        out.register_hook(lambda g: rec.__setitem__("dout", g.detach().clone()))
    return out
F.AttHeadFn.apply = staticmethod(wrap)
out = model.forward(img, q)
torch.nn.KLDivLoss()(out, soft).backward()
torch.cuda.synchronize()
F.AttHeadFn.apply = staticmethod(orig)

x, feat, w1, b1, wm, bm, w2, b2, unit, bf16, link = rec["args"]
dpooled = rec["dout"]
N, S, C = feat.shape
w1_2, w2_2 = w1.reshape(w1.shape[0], -1), w2.reshape(w2.shape[0], -1)
inv = link.inv
nrel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())

# kernel pieces
hid = ops.gemm_rowscale(x, w1_2, inv, S, bias=b1, relu=True)
logits, lin = ops.att_logits_fwd_lin(hid, w2_2, b2, b1)
wts, pooled = ops.glimpse_pool_fwd(feat, logits, False)
dlogits, _ = ops.glimpse_pool_bwd(dpooled.contiguous(), feat, wts, False, False)
d1s, dw2, db2, db1 = ops.att_logits_bwd(dlogits, hid, w2_2, relu_mask=True, rowscale=inv, rows_per_scale=S)
dx = ops.gemm(d1s, w1_2, tb=True)

# fp64 chain from the same operands
x64, w164, b164, w264, b264 = x.double(), w1_2.double(), b1.double(), w2_2.double(), b2.double()
sc = inv.double().repeat_interleave(S)[:, None]
pre64 = (x64 @ w164.t()) * sc + b164
hid64 = torch.relu(pre64)
print("hid vs fp64            %.2e" % nrel(hid, hid64))
logits64 = hid64 @ w264.t() + b264
print("logits vs fp64         %.2e" % nrel(logits, logits64))
wts64 = torch.softmax(logits64.view(N, S, -1), dim=1)            # (N,S,G)
print("wts vs fp64            %.2e" % nrel(wts, wts64.permute(0, 2, 1)))
dp64 = dpooled.double().view(N, -1, C)                            # (N,G,C)
dw64 = torch.einsum("ngc,nsc->nsg", dp64, feat.double())
dl64 = wts64 * (dw64 - (wts64 * dw64).sum(1, keepdim=True))
print("dlogits vs fp64        %.2e   (|dw| %.2e, |dw - dot| %.2e)" % (nrel(dlogits.view(N, S, -1), dl64), float(dw64.norm()),
                                                                     float((dw64 - (wts64 * dw64).sum(1, keepdim=True)).norm())))
# the same with the KERNEL's wts (isolates the backward arithmetic)
wk = wts.double().permute(0, 2, 1)
dl64k = wk * (dw64 - (wk * dw64).sum(1, keepdim=True))
print("dlogits vs fp64 (kernel's own wts) %.2e" % nrel(dlogits.view(N, S, -1), dl64k))
d1s64 = (dl64.reshape(N * S, -1) @ w264) * (hid64 > 0) * sc
print("d1s vs fp64            %.2e" % nrel(d1s, d1s64))
d1s64k = (dlogits.double() @ w264) * (hid.double() > 0) * sc
print("d1s vs fp64 of the kernel's dlogits %.2e" % nrel(d1s, d1s64k))
dx64 = d1s64 @ w164
print("dx vs fp64             %.2e" % nrel(dx, dx64))
print("dx vs fp64 product of the kernel's d1s %.2e" % nrel(dx, d1s.double() @ w164))
# torch fp32 on the same chain
dw32 = torch.einsum("ngc,nsc->nsg", dpooled.view(N, -1, C), feat)
w32 = torch.softmax((torch.relu((x @ w1_2.t()) * sc.float() + b1) @ w2_2.t() + b2).view(N, S, -1), dim=1)
dl32 = w32 * (dw32 - (w32 * dw32).sum(1, keepdim=True))
print("torch fp32 dlogits vs fp64 %.2e" % nrel(dl32, dl64))
