import sys, os
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tests/golden', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
sys.path[:0]=[os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests'), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests','golden')]
import torch, vqa_amd, recipe
from cases import MFB_CASES, MHBCOATT_CASES
from golden_util import recipe_sd, mfb_inputs
fn = vqa_amd.functions
mhb=True
case = dict(MHBCOATT_CASES[-1])
cfg, img, q, glove, hard, soft = mfb_inputs(case, "cuda")
model = vqa_amd.MHBCoAtt(cfg)
model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), case["salt"])) for k, v in model.state_dict().items()})
model = model.cuda().train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
cap={}
orig=fn.ImgFuseFn.apply
def wrapped(*a):
    Y=orig(*a)
    Y.register_hook(lambda g: cap.__setitem__('dY', g.detach().clone()))
    cap['Y']=Y.detach().clone()
    return Y
fn.ImgFuseFn.apply=wrapped
import importlib
mm=importlib.import_module('vqa-attention-networks_amd.host.mhb_coAtt')
class W:  # proxy so that the module picks up the wrapper
    apply=staticmethod(wrapped)
mm.ImgFuseFn=W
res={}
for mode in ("fp32","bf16-img","bf16-att","bf16"):
    model.gemm_dtype = mode
    model.zero_grad(set_to_none=True)
    out = model.forward(img, q)
    torch.nn.KLDivLoss()(out, soft).backward()
    res[mode]=dict(out=out.detach().clone(), Y=cap['Y'], dY=cap['dY'], g={k:p.grad.clone() for k,p in model.named_parameters()})
r0=res['fp32']
def rd(a,b): return float((a-b).norm()/(b.norm()+1e-30))
for mode in ("bf16-img","bf16-att","bf16"):
    r=res[mode]
    print(mode,'out',rd(r['out'],r0['out']),'Y',rd(r['Y'],r0['Y']),'dY',rd(r['dY'],r0['dY']), '|dY|',float(r0['dY'].norm()),
          ' grads:', {k: round(rd(r['g'][k],r0['g'][k]),4) for k in ('ques_proj1.weight','img_conv1d.weight','img_conv1d.bias','co_att_conv1.weight','ques_proj2.weight')})
# decomposition of dY relative to Y: cosine
Y=r0['Y']; dY=r0['dY']
N=case['N']; L=196
Yn=Y.view(N,-1); dYn=dY.view(N,-1)
print('cos(Y,dY) per sample', [float((Yn[i]*dYn[i]).sum()/(Yn[i].norm()*dYn[i].norm())) for i in range(N)])
