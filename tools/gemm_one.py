"""A few launches of ONE hot GEMM on live random operands, for rocprofv3 --pmc passes.
    python tools/gemm_one.py --dtype f32|bf16 --shape fwd|wgrad|hie_fwd|hie_dgrad|hie_wgrad|coatt_fwd|m512 [--out-bf16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32")
ap.add_argument("--shape", default="fwd")
ap.add_argument("--out-bf16", action="store_true")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
ops = vqa_amd.ops
vqa_amd.lib.load()
R = 512 * 196
g = torch.Generator().manual_seed(0)
if args.shape == "fwd":       # P = X W^T + b          (mfb.py:96)
    A = torch.relu(torch.randn((R, 2048), generator=g)).cuda()
    B = (torch.randn((5000, 2048), generator=g) * 0.03).cuda()
    ta = tb = False
    bias = torch.zeros(5000, device="cuda")
elif args.shape == "hie_fwd":  # HieCoAtten img_emb (hieCoAtten.py:25), B = 256: one round of 256x256 tiles + 17408 rows of 128x128 tiles
    A = torch.relu(torch.randn((256 * 196, 2048), generator=g)).cuda()
    B = (torch.randn((512, 2048), generator=g) * 0.03).cuda()
    ta = tb = False
    bias = torch.zeros(512, device="cuda")
elif args.shape == "m512":     # the M = 512 forward projections (mfb.py:126,127): one round of 128x80 tiles (csrc/gemm_f32_n80.hip)
    A = torch.randn((512, 2048), generator=g).cuda()
    B = (torch.randn((5000, 2048), generator=g) * 0.03).cuda()
    ta = tb = False
    bias = torch.zeros(5000, device="cuda")
elif args.shape == "hie_dgrad":  # d img = [dCv | dimg_] [Wbv; Wv]  (input gradient of hieCoAtten.py:30,35), B = 256, per-sample tiles
    A = torch.randn((256 * 196, 1024), generator=g).cuda()
    B = (torch.randn((1024, 512), generator=g) * 0.03).cuda()
    ta, tb = False, True
    bias = None
elif args.shape == "hie_wgrad":
    A = ((torch.rand((256 * 196, 512), generator=g) - 0.5) * 0.1).cuda()
    B = torch.relu(torch.randn((256 * 196, 2048), generator=g)).cuda()
    ta = tb = True
    bias = None
elif args.shape == "coatt_fwd":  # co_att_conv1 forward (mfb.py:109), K = 1000: 6 whole rounds + 2048 rows
    A = torch.randn((R, 1000), generator=g).cuda()
    B = (torch.randn((1024, 1000), generator=g) * 0.03).cuda()
    ta = tb = False
    bias = torch.zeros(1024, device="cuda")
else:                          # dW = dP^T X            (autograd of mfb.py:96), live dP
    A = ((torch.rand((R, 5000), generator=g) - 0.5) * 0.1).cuda()
    B = torch.relu(torch.randn((R, 2048), generator=g)).cuda()
    ta = tb = True
    bias = None
if args.dtype == "bf16":
    A, B = A.to(torch.bfloat16), B.to(torch.bfloat16)
    fn = lambda: ops.gemm_bf16(A, B, ta=ta, tb=tb, bias=bias, out_bf16=args.out_bf16)
elif args.shape in ("hie_fwd", "hie_dgrad"):      # per-sample tiles (csrc/gemm_f32_sample.hip) where the library takes the shape
    fn = lambda: ops.gemm_rows(A, B, 196, tb=tb, bias=bias)
else:
    fn = lambda: ops.gemm(A, B, ta=ta, tb=tb, bias=bias)
for _ in range(args.reps):
    fn()
torch.cuda.synchronize()
