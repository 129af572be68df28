"""Is the bf16-vs-fp32 gradient gap at B=512 (MHBCoAtt) a kernel error or the conditioning of the loss surface?
Three runs of the same model / batch: (a) fp32 kernels; (b) fp32 kernels on the image tensor rounded to bf16 and
widened back (a 4e-3 relative input perturbation, NO bf16 kernel involved); (c) gemm_dtype='bf16'.
If (b) moves the gradients as much as (c), the gap is the loss surface's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd, bench
vqa_amd.lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = bench.full_cfg("mhb_coAtt")
model = vqa_amd.MHBCoAtt(cfg)
bench.init_like_reference(model)
model = model.cuda().train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
img, q, a = bench.synth_batch(B, 0, "cuda")
soft = torch.softmax(torch.randn((B, 1000), generator=torch.Generator().manual_seed(7)), 1).cuda()
res = {}
for tag, mode, x in (("fp32", "fp32", img), ("fp32 on bf16-rounded img", "fp32", img.to(torch.bfloat16).float()), ("bf16 mode", "bf16", img)):
    model.gemm_dtype = mode
    model.zero_grad(set_to_none=True)
    out = model.forward(x, q)
    torch.nn.KLDivLoss()(out, soft).backward()
    torch.cuda.synchronize()
    res[tag] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
ref_o, ref_g = res["fp32"]
print("B = %d; relative deviation from the fp32 run (output, then per-tensor gradient norms)" % B)
for tag in ("fp32 on bf16-rounded img", "bf16 mode"):
    o, g = res[tag]
    print("%-26s out %.2e | " % (tag, float((o - ref_o).abs().max() / ref_o.abs().max())) +
          " ".join("%s=%.2f" % (k.replace(".weight", ".w").replace(".bias", ".b"), float((g[k] - ref_g[k]).norm() / (ref_g[k].norm() + 1e-30)))
                   for k in ref_g), flush=True)
