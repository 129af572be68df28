#!/usr/bin/env python3
"""A/B several builds of libvqa_fusion.so (tools/build_variant.sh) in ONE process on ONE device, interleaved
rounds, random operands, fp32 or bf16 GEMM entry point; prints the median per build and checks the results agree.
    python tools/gemm_libs_ab.py --dtype bf16 --shapes fwd,wgrad libA.so libB.so ..."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--shapes", default="fwd,wgrad")
ap.add_argument("--env", default="", help="KEY=VAL,KEY=VAL set before the launches")
args = ap.parse_args()
for kv in filter(None, args.env.split(",")):
    k, v = kv.split("=")
    os.environ[k] = v
SH = {"fwd": (0, 0, 100352, 5000, 2048), "wgrad": (1, 1, 5000, 2048, 100352), "coatt_fwd": (0, 0, 100352, 1024, 1024),
      "coatt_dgrad": (0, 1, 100352, 1024, 1024), "sq": (0, 0, 8192, 8192, 8192),
      "hie_fwd": (0, 0, 50176, 512, 2048), "hie_dgrad": (0, 1, 50176, 512, 1024), "hie_ci": (0, 0, 50176, 1024, 512),
      "coatt_wgrad": (1, 1, 1024, 1000, 100352), "coatt_wgrad512": (1, 1, 512, 1000, 100352)}
dev = torch.device("cuda")
bf = args.dtype == "bf16"
libs = []
for p in args.libs:
    l = ctypes.CDLL(os.path.abspath(p))
    fn = l.vqf_gemm_bf16 if bf else l.vqf_gemm_f32
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p]
    libs.append(fn)
ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for name in args.shapes.split(","):
    ta, tb, M, N, K = SH[name]
    g = torch.Generator(device="cpu").manual_seed(1)
    A = ((torch.rand((K, M) if ta else (M, K), generator=g) - 0.5) * 2).to(dev)
    B = ((torch.rand((K, N) if tb else (N, K), generator=g) - 0.5) * 2).to(dev)
    if bf:
        A, B = A.to(torch.bfloat16), B.to(torch.bfloat16)
    C = torch.empty((M, N), device=dev)

    def run(fn):
        rc = fn(ta, tb, M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0),
                None, 0, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    outs, times = [], [[] for _ in libs]
    for fn in libs:
        run(fn); torch.cuda.synchronize(); outs.append(C.clone())
    for r in range(args.rounds):
        for i, fn in enumerate(libs):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                run(fn)
            b.record(); torch.cuda.synchronize()
            times[i].append(a.elapsed_time(b) / 3)
    line = "%-11s (%d,%d) %s" % (name, ta, tb, args.dtype)
    for p, t, o in zip(args.libs, times, outs):
        t = sorted(t); med = t[len(t) // 2]
        d = float((o - outs[0]).abs().max() / outs[0].abs().max())
        line += " | %s %.3f ms %.0f TF d=%.0e" % (os.path.basename(p).replace("libvqf_", "").replace(".so", ""), med,
                                                  2.0 * M * N * K / med / 1e9, d)
    print(line, flush=True)
    del A, B, C, outs
