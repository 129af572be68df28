"""Per-step cost of the LSTM recursion kernels alone: per-step launches vs the persistent launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
S, B, H = int(os.environ.get("S", 512)), int(os.environ.get("B", 14)), int(os.environ.get("H", 1024))
g = torch.Generator().manual_seed(0)
xw = ((torch.rand((S, B, 4 * H), generator=g) * 2 - 1) * 1.5).cuda()
w_hh = ((torch.rand((4 * H, H), generator=g) * 2 - 1) * 0.04).cuda()
dhs = (torch.rand((S, B, H), generator=g) * 2 - 1).cuda()
w_t = w_hh.t().contiguous()


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


hs, cs, gt = ops.lstm_seq_fwd(xw, w_hh)
for name, f in (("fwd per-step", lambda: ops.lstm_seq_fwd(xw, w_hh)),
                ("fwd persist ", lambda: ops.lstm_seq_fwd_persist(xw, w_hh)),
                ("bwd per-step", lambda: ops.lstm_seq_bwd(dhs, gt, cs, w_hh)),
                ("bwd persist ", lambda: ops.lstm_seq_bwd_persist(dhs, gt, cs, w_hh))):
    ms = timed(f)
    print("%s  %.3f ms  = %.2f us/step" % (name, ms, ms * 1e3 / S), flush=True)
ops.lstm_persist_status()
