#!/usr/bin/env python3
"""Micro-benchmark of the dominant GEMM launches (bench shapes), for kernel tuning and PMC passes.

    python tools/gemm_bench.py [--iters 5] [--which fwd,wgrad,coatt]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vqa_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--which", default="fwd,wgrad,coatt_fwd,coatt_dgrad,small")
ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()
ops = vqa_amd.ops
vqa_amd.lib.load()
dev = torch.device("cuda")
M = args.batch * 196
g = torch.Generator(device="cpu").manual_seed(0)


def rnd(*shape):
    return (torch.rand(shape, generator=g) * 2 - 1).to(dev)


def timeit(name, fn, flops):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    med = ts[len(ts) // 2]
    print("%-28s %9.3f ms (min %.3f)  %7.1f TFLOP/s  %.3f of fp32 MFMA peak" %
          (name, med, ts[0], flops / med / 1e9, flops / med / 1e9 / 157.3), flush=True)


which = args.which.split(",")
if "fwd" in which:
    X, W, b = torch.relu(rnd(M, 2048)), rnd(5000, 2048) * 0.03, rnd(5000)
    out = torch.empty(M, 5000, device=dev)
    timeit("img_conv1d fwd  (a0b0)", lambda: ops.gemm(X, W, bias=b, out=out), 2.0 * M * 5000 * 2048)
    del out
if "wgrad" in which:
    X, dP = torch.relu(rnd(M, 2048)), rnd(M, 5000)
    timeit("img_conv1d wgrad (a1b1)", lambda: ops.gemm(dP, X, ta=True, tb=True), 2.0 * M * 5000 * 2048)
    del dP
if "coatt_fwd" in which:
    Y, W = rnd(M, 1000), rnd(1024, 1000)
    timeit("co_att_conv1 fwd (a0b0)", lambda: ops.gemm(Y, W, relu=True), 2.0 * M * 1024 * 1000)
if "coatt_dgrad" in which:
    dH, W = rnd(M, 1024), rnd(1024, 1000)
    timeit("co_att_conv1 dgrad (a0b1)", lambda: ops.gemm(dH, W, tb=True), 2.0 * M * 1024 * 1000)
    Y = rnd(M, 1000)
    timeit("co_att_conv1 wgrad (a1b1)", lambda: ops.gemm(dH, Y, ta=True, tb=True), 2.0 * M * 1024 * 1000)
if "small" in which:
    A, W = rnd(args.batch, 2048), rnd(5000, 2048)
    timeit("ques_proj fwd M=512", lambda: ops.gemm(A, W), 2.0 * args.batch * 5000 * 2048)
    A4, W4 = rnd(args.batch, 4096), rnd(5000, 4096)
    timeit("img_proj2 fwd M=512 K=4096", lambda: ops.gemm(A4, W4), 2.0 * args.batch * 5000 * 4096)
    dQ = rnd(args.batch, 5000)
    timeit("img_proj2 dgrad", lambda: ops.gemm(dQ, W4, tb=True), 2.0 * args.batch * 5000 * 4096)
    timeit("img_proj2 wgrad", lambda: ops.gemm(dQ, A4, ta=True, tb=True), 2.0 * args.batch * 5000 * 4096)

if "bf16" in which:
    Xb = ops.cast_bf16(torch.relu(rnd(M, 2048)))
    Wb = ops.cast_bf16(rnd(5000, 2048) * 0.03)
    b = rnd(5000)
    out = torch.empty(M, 5000, device=dev)
    timeit("bf16 img_conv1d fwd (a0b0)", lambda: ops.gemm_bf16(Xb, Wb, bias=b, out=out), 2.0 * M * 5000 * 2048)
    dPb = ops.cast_bf16(rnd(M, 5000))
    timeit("bf16 img_conv1d wgrad (a1b1)", lambda: ops.gemm_bf16(dPb, Xb, ta=True, tb=True), 2.0 * M * 5000 * 2048)
    Yb, W1b = ops.cast_bf16(rnd(M, 1000), 32), ops.cast_bf16(rnd(1024, 1000), 32)
    timeit("bf16 co_att_conv1 fwd", lambda: ops.gemm_bf16(Yb, W1b, K=1024, relu=True), 2.0 * M * 1024 * 1000)
    dHb = ops.cast_bf16(rnd(M, 1024))
    timeit("bf16 co_att_conv1 dgrad", lambda: ops.gemm_bf16(dHb, W1b, tb=True, N=1000), 2.0 * M * 1024 * 1000)
    timeit("bf16 co_att_conv1 wgrad", lambda: ops.gemm_bf16(dHb, Yb, ta=True, tb=True), 2.0 * M * 1024 * 1000)
    x32 = rnd(M, 5000)
    timeit("cast f32->bf16 (2 GB)", lambda: ops.cast_bf16(x32), 1.0)
