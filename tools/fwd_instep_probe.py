"""Why is the image projection's forward launch ~2 % slower inside the train step than alone?  One process, one box:
the same vqf_gemm_f32 launch (M=100352, N=5000, K=2048) under the conditions the step adds, one at a time.
    python tools/fwd_instep_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
dev = torch.device("cuda")
M, N, K = 100352, 5000, 2048
g = torch.Generator(device="cuda").manual_seed(5)
A_uni = (torch.rand((M, K), device=dev, generator=g) - 0.5) * 2
A_relu = torch.relu(torch.randn((M, K), device=dev, generator=g))
W_uni = (torch.rand((N, K), device=dev, generator=g) - 0.5) * 2
W_xav = torch.empty((N, K), device=dev)
torch.nn.init.xavier_uniform_(W_xav)
bias = torch.zeros(N, device=dev)
C = torch.empty((M, N), device=dev)


def timed(fn, reps=5, pre=None):
    ts = []
    for _ in range(reps):
        if pre is not None:
            pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


small = torch.empty((512, 1000), device=dev)
big_other = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
h512 = torch.randn((512, 1024), device=dev); whh = torch.randn((4096, 1024), device=dev); g512 = torch.empty((512, 4096), device=dev)
q512 = torch.randn((512, 2048), device=dev); wq = torch.randn((5000, 2048), device=dev); p512 = torch.empty((512, 5000), device=dev)
big_src = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
cases = [
    ("uniform A, uniform W, no bias, out reused", lambda: ops.gemm(A_uni, W_uni, out=C), None),
    ("+ bias", lambda: ops.gemm(A_uni, W_uni, bias=bias, out=C), None),
    ("relu(randn) A, uniform W", lambda: ops.gemm(A_relu, W_uni, bias=bias, out=C), None),
    ("relu(randn) A, xavier W (the step's operands)", lambda: ops.gemm(A_relu, W_xav, bias=bias, out=C), None),
    ("... fresh output tensor each launch", lambda: ops.gemm(A_relu, W_xav, bias=bias), None),
    ("... after 40 small launches", lambda: ops.gemm(A_relu, W_xav, bias=bias, out=C),
     lambda: [small.mul_(1.0) for _ in range(40)]),
    ("... after a 512 MB memset (caches dirtied)", lambda: ops.gemm(A_relu, W_xav, bias=bias, out=C), lambda: big_other.zero_()),
    ("... after an idle 30 ms (host sleep)", lambda: ops.gemm(A_relu, W_xav, bias=bias, out=C),
     lambda: (torch.cuda.synchronize(), __import__("time").sleep(0.03))),
    ("... after 14 recurrent-size products (512x4096x1024) + 10 M=512 projections (~2.3 ms of small GEMMs)",
     lambda: ops.gemm(A_relu, W_xav, bias=bias, out=C),
     lambda: ([ops.gemm(h512, whh, out=g512) for _ in range(14)], [ops.gemm(q512, wq, out=p512) for _ in range(10)])),
    ("... after 3 ms of HBM-bound copies", lambda: ops.gemm(A_relu, W_xav, bias=bias, out=C),
     lambda: [big_other.copy_(big_src) for _ in range(12)]),
    ("... after the Adam-sized elementwise pass + 1 ms gap of tiny launches", lambda: ops.gemm(A_relu, W_xav, bias=bias, out=C),
     lambda: [small.mul_(1.0) for _ in range(200)]),
    ("uniform again", lambda: ops.gemm(A_uni, W_uni, out=C), None),
]
for name, fn, pre in cases:
    fn(); torch.cuda.synchronize()
    ms = timed(fn, pre=pre)
    print("%-55s %8.3f ms  %6.1f TF" % (name, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
