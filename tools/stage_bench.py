"""Input staging measurements: transpose kernel vs HBM roofline, H2D rate, and
the MFB train step fed from pinned host memory through FeatureStager vs an HBM-resident batch.

    python tools/stage_bench.py [--batch 512] [--steps 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd  # noqa: E402
import bench as B  # noqa: E402


def timed(fn, iters):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=8)
    args = ap.parse_args()
    N, D, L = args.batch, 2048, 196
    dev = torch.device("cuda", 0)
    ops = vqa_amd.ops
    out = {}

    raw = torch.randn((N, D, L), device=dev).relu_()
    for bf in (False, True):
        ops.feat_transpose(raw, bf16=bf)
        ms = timed(lambda: ops.feat_transpose(raw, bf16=bf), 20)
        nbytes = N * D * L * (4 + (2 if bf else 4))
        out["transpose_%s" % ("bf16" if bf else "f32")] = {
            "ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 3)}

    host = torch.empty((N, D, L)).pin_memory()
    host.copy_(raw.cpu())
    dst = torch.empty_like(raw)
    dst.copy_(host, non_blocking=True)
    ms = timed(lambda: dst.copy_(host, non_blocking=True), 5)
    out["h2d_pinned"] = {"ms": round(ms, 3), "GBps": round(N * D * L * 4 / ms / 1e6, 1)}

    # train step: resident batch vs staged batch (double-buffered)
    cfg = B.full_cfg("mfb")
    model = vqa_amd.MFB(cfg)
    B.init_like_reference(model)
    model = model.to(dev).train()
    opt = vqa_amd.Adam(model.parameters(), lr=7e-4)
    crit = vqa_amd.CrossEntropyLoss()
    img, q, a = B.synth_batch(N, 0, dev)

    def step(x):
        opt.zero_grad(set_to_none=True)
        loss = crit(model.forward(x, q), a)
        loss.backward()
        opt.step()

    for _ in range(2):
        step(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(img)
    torch.cuda.synchronize()
    resident = (time.perf_counter() - t0) / args.steps * 1e3

    st = vqa_amd.FeatureStager(N, D, L, device=dev, depth=2)
    src = host.numpy()
    def fill():
        h = st.host_slot()
        np.copyto(h, src)          # stands for np.load into the pinned slot
        st.commit()
    fill()
    step(st.next())
    fill()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fill_s = 0.0
    for _ in range(args.steps):
        x = st.next()
        step(x)                    # GPU works on batch k ...
        f0 = time.perf_counter()
        fill()                     # ... while the host fills and ships batch k+1
        fill_s += time.perf_counter() - f0
    torch.cuda.synchronize()
    staged = (time.perf_counter() - t0) / args.steps * 1e3
    out["train_step_ms"] = {"resident": round(resident, 2), "staged_pcie_inclusive": round(staged, 2),
                            "host_fill_ms": round(fill_s / args.steps * 1e3, 2),
                            "qa_per_s_resident": round(N / resident * 1e3, 1),
                            "qa_per_s_staged": round(N / staged * 1e3, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
