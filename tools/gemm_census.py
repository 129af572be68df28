"""Every GEMM launch of one train step: layout, shape, time (events around each call, synchronised: launch gaps do
not count), split-K reduce launches; grouped by shape.   python tools/gemm_census.py [--model mfb] [--dtype f32]"""
import argparse, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="mfb")
ap.add_argument("--dtype", default="f32")
args = ap.parse_args()
ops = vqa_amd.ops
vqa_amd.lib.load()
HIE = args.model == "hieCoAtten"
B = 256 if HIE else 512
if HIE:
    model = vqa_amd.HieCoAtten(block_num=196, word_num=14, img_size=2048, vocab_size=1000, embed_size=512,
                               output_size=1000)
else:
    model = (vqa_amd.MFB if args.model == "mfb" else vqa_amd.MHBCoAtt)(bench.full_cfg(args.model))
bench.init_like_reference(model)
model = model.cuda().train()
if not HIE:
    model.gemm_dtype = "bf16" if args.dtype == "bf16" else "fp32"
    model.overlap_streams = "same-stream"
opt = vqa_amd.Adam(model.parameters(), lr=7e-4)
crit = vqa_amd.train_step.criterion_for(args.model)
img, q, a = bench.synth_batch(B, 0, "cuda")
soft = torch.softmax(torch.randn((B, 1000)), 1).cuda()
if args.dtype == "bf16":
    img = ops.cast_bf16(img.view(-1, 2048)).view(img.shape)


def step():
    opt.zero_grad(set_to_none=True)
    out = model.forward(img, q)
    loss = crit(out[0] if HIE else out, soft if args.model == "mhb_coAtt" else a)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
rec = collections.OrderedDict()


def wrap(name, real):
    def f(a_, b_, ta=False, tb=False, **kw):
        M = kw.get("M") or (a_.shape[1] if ta else a_.shape[0])
        K = kw.get("K") or (a_.shape[0] if ta else a_.shape[1])
        N = kw.get("N") or (b_.shape[1] if tb else b_.shape[0])
        torch.cuda.synchronize()
        ops.prof_reset(); ops.prof_enable(True)
        out = real(a_, b_, ta=ta, tb=tb, **kw)
        torch.cuda.synchronize()
        ops.prof_enable(False)
        rep = ops.prof_report()
        red = rep.pop("splitk_reduce", (0, 0.0))
        ms = sum(v[1] for v in rep.values())
        key = (name, int(bool(ta)), int(bool(tb)), M, N, K, bool(kw.get("accumulate")))
        r = rec.setdefault(key, [0, 0.0, 0, 0.0])
        r[0] += 1; r[1] += ms; r[2] += red[0]; r[3] += red[1]
        return out
    return f


def wrap_b(real):
    def f(a_, b_, ta=False, tb=False, **kw):
        Bn = a_.shape[0]
        M = a_.shape[2] if ta else a_.shape[1]
        K = a_.shape[1] if ta else a_.shape[2]
        N = b_.shape[2] if tb else b_.shape[1]
        torch.cuda.synchronize()
        ops.prof_reset(); ops.prof_enable(True)
        out = real(a_, b_, ta=ta, tb=tb, **kw)
        torch.cuda.synchronize()
        ops.prof_enable(False)
        ms = sum(v[1] for v in ops.prof_report().values())
        key = ("b%d" % Bn, int(bool(ta)), int(bool(tb)), M, N, K, bool(kw.get("accumulate")))
        r = rec.setdefault(key, [0, 0.0, 0, 0.0])
        r[0] += 1; r[1] += ms
        return out
    return f


ops.gemm, ops.gemm_bf16, ops.bgemm = wrap("f32", ops.gemm), wrap("bf16", ops.gemm_bf16), wrap_b(ops.bgemm)
step()
torch.cuda.synchronize()
tot = 0.0
print("%-5s %-5s %7s %6s %7s %4s %6s %9s %8s %7s %9s" % ("type", "ta,tb", "M", "N", "K", "acc", "calls", "ms/call", "TF", "reduces", "red ms"))
for (name, ta, tb, M, N, K, acc), (n, ms, nr, rms) in sorted(rec.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    nb = int(name[1:]) if name.startswith("b") and name[1:].isdigit() else 1
    print("%-5s (%d,%d) %7d %6d %7d %4s %6d %9.4f %8.1f %7d %9.4f" % (name, ta, tb, M, N, K, "acc" if acc else "", n, ms / n,
                                                                     2.0 * nb * M * N * K * n / ms / 1e9, nr, rms))
    tot += ms + rms
print("total GEMM + reduce time per step: %.3f ms" % tot)
