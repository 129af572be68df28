"""Where a ping-pong GEMM wave spends its cycles: run with a diagnostic build that stamps the segments of the
K loop with s_memtime (never the product build, never quote its run time):
    tools/build_variant.sh stamps "-DVQF_PP_STAMPS" gemm_bf16_big.hip
    VQF_LIB=variants/libvqf_stamps.so python tools/pp_stamps.py
Segments per slab and wave (mean shader cycles): reads = 12 fragment reads issued AND returned; copies = issue of the
4 LDS-DMA refills; vmcnt = wait for slab s+1; bar1 = lgkmcnt + barrier that ends L; mfma = issue of the 16 MFMAs;
bar2 = barrier that ends M; loop = loop overhead between the slabs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
for name, ta, tb, M, N, K in (("img fwd (0,0)", 0, 0, 100352, 5000, 2048), ("sq 8192", 0, 0, 8192, 8192, 8192)):
    A = (torch.rand((K, M) if ta else (M, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((K, N) if tb else (N, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_bf16(A, B, ta=bool(ta), tb=bool(tb), out=out, out_bf16=True)
    torch.cuda.synchronize()
    ws = ops.workspace(A.device, ops.SPLITK_WS_BYTES)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    d = ws[:tiles * 8 * 8 * 8].view(torch.int64).view(tiles, 8, 8).cpu().numpy().astype(np.float64)
    S = d[:, :, 7:8]
    per = d[:, :, :7] / S
    names = ["reads", "copies", "vmcnt", "bar1", "mfma", "bar2", "loop"]
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        m = per[:, sl, :].mean(axis=(0, 1))
        print("%-14s %-10s " % (name, grp) + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, m)) +
              "  | total %.0f cycles per slab (512 = MFMA-bound)" % m.sum(), flush=True)
