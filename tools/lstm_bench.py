import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
vqa_amd.lib.load()
fn = vqa_amd.functions.LstmSeqFn
S, B, I, H = 512, 14, 300, 1024
torch.manual_seed(0)
x = torch.randn(S, B, I, device="cuda", requires_grad=True)
ps = [torch.randn(4 * H, I, device="cuda") * 0.05, torch.randn(4 * H, H, device="cuda") * 0.03,
      torch.zeros(4 * H, device="cuda"), torch.zeros(4 * H, device="cuda")]
for p in ps:
    p.requires_grad_()
for it in range(3):
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.record()
    hs = fn.apply(x, *ps)
    b.record()
    hs.sum().backward()
    c.record()
    torch.cuda.synchronize()
    print("fwd %.2f ms  bwd %.2f ms" % (a.elapsed_time(b), b.elapsed_time(c)), flush=True)
