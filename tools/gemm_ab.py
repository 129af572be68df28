#!/usr/bin/env python3
"""A/B two builds of libvqa_fusion.so in ONE process on ONE device (interleaved rounds), because
device-to-device and run-to-run spread (a few %) exceeds the deltas being tuned.

    python tools/gemm_ab.py libA.so libB.so [--rounds 7] [--shape fwd|wgrad|coatt]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--shape", default="fwd")
args = ap.parse_args()

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.rand(s, generator=g) * 2 - 1).to(dev)
B = 512
M = B * 196
if args.shape == "fwd":
    ta, tb, m, n, k = 0, 0, M, 5000, 2048
    A, Bm = torch.relu(rnd(M, 2048)), rnd(5000, 2048) * 0.03
elif args.shape == "wgrad":
    ta, tb, m, n, k = 1, 1, 5000, 2048, M
    A, Bm = rnd(M, 5000), torch.relu(rnd(M, 2048))
else:
    ta, tb, m, n, k = 0, 0, M, 1024, 1000
    A, Bm = rnd(M, 1000), rnd(1024, 1000)
C = torch.empty(m, n, device=dev)
ws = torch.empty(400 << 20, dtype=torch.uint8, device=dev)
bias = rnd(n)
libs = []
for p in args.libs:
    l = ctypes.CDLL(os.path.abspath(p))
    l.vqf_gemm_f32.restype = ctypes.c_int
    l.vqf_gemm_f32.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                    ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                    ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    libs.append(l)


def run(l):
    rc = l.vqf_gemm_f32(ta, tb, m, n, k, A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0), C.data_ptr(),
                        C.stride(0), bias.data_ptr(), 0, ws.data_ptr(), ws.numel(),
                        torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


times = [[] for _ in libs]
for l in libs:
    run(l)
torch.cuda.synchronize()
for r in range(args.rounds):
    for i, l in enumerate(libs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run(l)
        run(l)
        b.record()
        torch.cuda.synchronize()
        times[i].append(a.elapsed_time(b) / 2)
outs = []
for l in libs:
    C.zero_()
    run(l)
    torch.cuda.synchronize()
    outs.append(C.clone())
for p, o in zip(args.libs[1:], outs[1:]):
    d = float((o - outs[0]).abs().max() / outs[0].abs().max())
    print("%s vs %s: max rel diff %.2e" % (os.path.basename(p), os.path.basename(args.libs[0]), d), flush=True)
fl = 2.0 * m * n * k
for p, t in zip(args.libs, times):
    t = sorted(t)
    print("%-40s median %.3f ms (min %.3f max %.3f)  %.1f TF" % (os.path.basename(p), t[len(t) // 2], t[0], t[-1],
                                                                 fl / t[len(t) // 2] / 1e9), flush=True)
