#!/usr/bin/env python3
"""Headline benchmark: QA-pairs/s of one MFB-baseline training step (forward +
loss + backward + gradient all-reduce + Adam step), batch 512 per GPU, fp32,
on 1..8 MI355X (one process per GPU, RCCL over xGMI), beside the CPU baseline.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  Synthetic
inputs per BASELINE.md section 3: img = relu(N(0,1)) (B,196,2048), questions
uniform [1,1000) (B,14), labels uniform [0,1000); weights: manual_seed(0),
module default init, xavier_uniform_ on every non-bias parameter
(train_models.py:54-56).  Train mode (dropout active), device-resident inputs.

mode "faithful": every op the reference's autograd executes is executed here
too -- including the image-projection GEMM and its weight-gradient GEMM whose
results cannot reach the logits/gradients in MFB-baseline because mfb.py:84,118
take their softmax over a singleton axis.  roofline.achieved
counts only FLOPs actually executed by the measured kernel.

After the headline (BASELINE config 2) a default 1-GPU run frees the model and
times, in the same process, BASELINE config 3 (MHBCoAtt, bf16 operands, B=512)
and config 4 (HieCoAtten, fp32, B=256): `secondary.config3 / .config4`, each
with its own ms_per_step, value, dtype and roofline object.  The headline keys
are unaffected (`--no-secondary` skips them).
"""
import argparse
import glob
import json
import os
import sys
import time
import types

_T0 = time.time()        # process start, before `import torch` (1-2 minutes on a fresh box): the stage markers count from here

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak (= vector peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 matrix peak (no sparsity)
HBM_PEAK_GBS = 8000.0
# algorithmic GEMM FLOPs of one train step per QA pair (SURVEY 8d / BASELINE.md section 4), forward + backward
STEP_MFLOP_PER_QA = {"mfb": 10034.0, "mhb_coAtt": 9578.0, "hieCoAtten": 1515.0}
CENSUS_STEPS = 4
C3_SIDE_CUS = 128
KERNELS_NOTE = ("per-kernel times from %d further UNTIMED steps with every library kernel bracketed by hipEvents; the timed "
                "region brackets only the dominant GEMM launches (a bracket costs the stream ~6-10 us)" % CENSUS_STEPS)


def full_cfg(model_name="mfb"):
    return types.SimpleNamespace(q_vocab_size=1000, a_vocab_size=1000, emb_dim=300, hidden_dim=1024,
                                 num_layers=1, model_name=model_name, glove=False,
                                 img_feature_channel=2048, img_feature_dim=196)


def synth_batch(B, rank, device):
    g = torch.Generator().manual_seed(1234 + 1000 * rank)
    img = torch.relu(torch.randn((B, 196, 2048), generator=g))
    q = torch.randint(1, 1000, (B, 14), generator=torch.Generator().manual_seed(1235 + 1000 * rank))
    a = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(1236 + 1000 * rank))
    return img.to(device), q.to(device), a.to(device)


def init_like_reference(model):
    torch.manual_seed(0)
    for name, p in model.named_parameters():
        if name.find('bias') == -1:
            torch.nn.init.xavier_uniform_(p)          # train_models.py:54-56


def host_cores():
    """CPU cores this process may actually use: cgroup quota (cpu.max) capped by the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(batch=512, warmups=2, timed=5, config1_warmups=3, config1_timed=5):
    """The oracle (CPU restatement, PyTorch CPU ops) on the host cores, on the metric's own configuration
    (BASELINE.md section 3): the MFB train step (fwd + loss + bwd + Adam) at B=512, `warmups` warm-ups + `timed`
    steps, median; plus BASELINE config 1 (MFB forward, eval, B=32) as `config1`.  The dropout keep-masks are drawn
    per step like the reference's nn.Dropout does (33 % of the reference's CPU step); `value` is the
    CONSERVATIVE rate with the mask draws outside the timed region, `with_mask_generation` the rate with them inside.
    About 75 s on the 16 cores of a GPU box; `--cpu-batch` shrinks the sample for quick runs."""
    from oracle import ref_torch as O
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = full_cfg()
    torch.manual_seed(0)
    sd = {}
    for k, shp in O.mfb_shapes(cfg).items():
        t = torch.empty(shp)
        if k.find('bias') == -1 and t.dim() >= 2:
            torch.nn.init.xavier_uniform_(t)
        else:
            t.uniform_(-0.05, 0.05)
        sd[k] = t.requires_grad_(True)
    opt = torch.optim.Adam(list(sd.values()), lr=7e-4)
    gen = torch.Generator().manual_seed(99)

    def one_step(img, q, a):
        Bs = img.shape[0]
        tm = time.perf_counter()
        drop = dict(l=(torch.rand((Bs, 14, 1024), generator=gen) >= 0.3),
                    m1=(torch.rand((Bs, 196, 5000), generator=gen) >= 0.1),
                    m2=(torch.rand((Bs, 5000), generator=gen) >= 0.1))
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        logits = O.mfb_forward(sd, cfg, img, q, drop=drop)
        loss = O.ce_loss(logits, a)
        loss.backward()
        opt.step()
        t1 = time.perf_counter()
        return t1 - t0, t1 - tm

    def fwd_only(img, q):
        t0 = time.perf_counter()
        with torch.no_grad():
            O.mfb_forward(sd, cfg, img, q)              # eval: no dropout (BASELINE config 1)
        return time.perf_counter() - t0

    img, q, a = synth_batch(batch, 0, "cpu")
    one_step(img[:16], q[:16], a[:16])                      # page-in / thread-pool warm-up
    for _ in range(warmups):
        one_step(img, q, a)                                 # warm-ups at the full batch
    runs = [one_step(img, q, a) for _ in range(timed)]
    t = sorted(r[0] for r in runs)[timed // 2]
    tmask = sorted(r[1] for r in runs)[timed // 2]
    for _ in range(config1_warmups):
        fwd_only(img[:32], q[:32])
    t1s = sorted(fwd_only(img[:32], q[:32]) for _ in range(config1_timed))
    t1 = t1s[len(t1s) // 2]
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": round(batch / t, 3), "unit": "QA-pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle MFB train step (fwd+loss+bwd+Adam; dropout keep-masks drawn per step OUTSIDE the timed "
                      "region), B=%d, %d warm-ups + %d timed steps, median %.2f s/step, %d threads; CPU: %s"
                      % (batch, warmups, timed, t, cores, model),
            "with_mask_generation": {"value": round(batch / tmask, 3), "unit": "QA-pairs/s",
                                     "sample": "the same steps with the three keep-mask draws (torch.rand >= p; the reference "
                                               "draws them inside nn.Dropout) inside the timed region, median %.2f s/step" % tmask},
            "config1": {"value": round(32 / t1, 3), "unit": "QA-pairs/s",
                        "sample": "BASELINE config 1: oracle MFB forward, eval, B=32, %d warm-ups + %d timed, "
                                  "median %.3f s, %d threads" % (config1_warmups, config1_timed, t1, cores)}}


def live_wgrad_probe(ops, B, dev, reps=6):
    """The weight-gradient launch of the image projection on LIVE operands.  In faithful MFB the singleton-axis
    softmax makes dP exactly zero, and MFMA loops hold a higher clock on zeros (MI355X_MICROARCH.md, DVFS
    give-back), so the in-step timing of that launch is no evidence for real data.  Same shape, same entry
    point (ops.gemm(dP, X, ta, tb)), uniform random dP and a relu(N(0,1)) image; hipEvent-timed by the library's
    profiler on the launch stream.  Returns (launches, total_ms, splitk_reduce_ms)."""
    M, N, K = 5000, 2048, B * 196
    g = torch.Generator(device="cpu").manual_seed(4321)
    X = torch.relu(torch.randn((K, N), generator=g)).to(dev)
    dP = ((torch.rand((K, M), generator=g) - 0.5) * 0.1).to(dev)
    for _ in range(2):
        ops.gemm(dP, X, ta=True, tb=True)
    torch.cuda.synchronize()
    ops.prof_reset()
    ops.prof_enable(True)
    for _ in range(reps):
        ops.gemm(dP, X, ta=True, tb=True)
    torch.cuda.synchronize()
    ops.prof_enable(False)
    n, ms = ops.prof_gemm(1, 1, M, N, K)
    red = ops.prof_report().get("splitk_reduce", (0, 0.0))[1]
    del X, dP
    return n, ms, red


def hbm_yardsticks(ops, B, dev, reps=6):
    """What this chip reaches on plain streaming kernels over a tensor of the projection's size ((B*196, 5000) fp32, 2 GB --
    far beyond the 256 MiB Infinity Cache), `reps` launches back to back between two events on the current stream, bytes
    read + written / time.  The library's own 16-B-per-lane grid-stride kernels (csrc/yardstick.hip): `copy` (read : write =
    1 : 1, the mix of mfb_fuse_bwd) and `read_sweep` (a pure read stream: the mix of glimpse_pool_fwd, 5 : 1 for mfb_fuse_fwd),
    each with default-policy and non-temporal accesses; MI355X_MICROARCH.md measures 6.29 TB/s (float4 copy) and 6.0-6.1 TB/s
    (read sweep).  `torch_copy` (round 3's yardstick, torch's copy kernel) is kept for comparison.  The HBM-bound kernels of
    the step are priced against the 8 TB/s pin rate; these say what a pure stream gets of it on this box."""
    src = torch.empty((B * 196, 5000), dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    nb = float(src.numel() * 4)
    sums = ops.hbm_read_sweep(src)

    def timed(fn, nbytes, what):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"what": what, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "ms": round(ms, 4)}
    out = {
        "copy": timed(lambda: ops.hbm_copy(src, dst), 2 * nb, "vqf_hbm_copy, 2 GB read + 2 GB written"),
        "copy_nt": timed(lambda: ops.hbm_copy(src, dst, nt=True), 2 * nb, "the same with non-temporal loads / stores"),
        "read_sweep": timed(lambda: ops.hbm_read_sweep(src, out=sums), nb, "vqf_hbm_read_sweep, 2 GB read"),
        "read_sweep_nt": timed(lambda: ops.hbm_read_sweep(src, nt=True, out=sums), nb, "the same with non-temporal loads"),
        "torch_copy": timed(lambda: dst.copy_(src), 2 * nb, "torch.Tensor.copy_ (round 3's yardstick)"),
    }
    assert torch.equal(dst, src)
    del src, dst
    return out


def standalone_bf16_projection(ops, B, dev, reps=6):
    """The bf16 image-projection launch ALONE on the chip (all CUs, nothing beside it), live operands, bf16 output: what the
    kernel itself reaches when config 3's step does not make it share the chip with the LSTM recursion."""
    M, N, K = B * 196, 5000, 2048
    g = torch.Generator(device="cpu").manual_seed(4322)
    X = torch.relu(torch.randn((M, K), generator=g)).to(dev).to(torch.bfloat16)
    W = ((torch.rand((N, K), generator=g) - 0.5) * 0.06).to(dev).to(torch.bfloat16)
    for _ in range(2):
        ops.gemm_bf16(X, W, out_bf16=True)
    torch.cuda.synchronize()
    ops.prof_reset()
    ops.prof_enable(True, min_mnk=M * N * K // 2)
    for _ in range(reps):
        ops.gemm_bf16(X, W, out_bf16=True)
    torch.cuda.synchronize()
    ops.prof_enable(False)
    n, ms = ops.prof_shape("gemm_bf16", M, N, K)
    ops.prof_reset()
    del X, W
    if not n:
        return None
    ach = 2.0 * M * N * K / (ms / n * 1e-3) / 1e12
    return {"achieved": round(ach, 2), "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms / n, 4), "launches": n,
            "what": "the same launch alone on all 256 CUs, outside the step, %d launches back to back (sustained: the chip lowers its "
                    "clock under continuous bf16 MFMA load; once per step in the one-stream form of this step it takes 1.68 ms = 0.49)" % reps}


_PMC_EXTRA = {}      # (dtype, M, N, K) -> {clock_ghz, mfma_busy_frac, l2_hit_rate} of the committed counter pass, filled by pmc_lookup


def pmc_lookup(dtype, M, N, K):
    """Beyond-L2 bytes per launch of a GEMM from the committed rocprofv3 --pmc passes (PMC passes cannot run inside
    the timed process): newest profiles/r*_pmc_*.json whose dtype / M / N / K match.  -> (bytes, note, source) or Nones."""
    for src in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")), reverse=True):
        try:
            recs = json.load(open(src))
        except Exception:
            continue
        for pmc in (recs if isinstance(recs, list) else [recs]):
            try:
                if pmc.get("dtype", "f32") == dtype and (pmc["M"], pmc["N"], pmc["K"]) == (M, N, K):
                    _PMC_EXTRA[(dtype, M, N, K)] = {k: pmc[k] for k in ("clock_ghz", "mfma_busy_frac", "l2_hit_rate") if k in pmc}
                    return pmc["traffic_bytes"], pmc.get("note", "") + "; " + pmc.get("formula", ""), os.path.relpath(src, ROOT)
            except Exception:
                continue
    return None, None, None


def pmc_kernel_lookup(name):
    """The same for the HBM-bound kernels: profiles/r*_pmc_fuse.json holds {kernel: {traffic_bytes, algorithmic_bytes, ...}}."""
    for src in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fuse.json")), reverse=True):
        try:
            rec = json.load(open(src)).get(name)
            if rec:
                return rec, os.path.relpath(src, ROOT)
        except Exception:
            continue
    return None, None


def gemm_roofline(ops, kernel_id, dtype, M, N, K, what, peak, ta=0, tb=0, sample_rows=0):
    """roofline object of one GEMM of the step, timed by the library's hipEvent profiler (shape-tagged launches).  A mid-size
    fp32 product is TWO launches (vqf_gemm_f32_big_rows: whole rounds of the 256x256-tile kernel + the remaining rows on the
    128x128 kernel): the object then covers both -- all of the product's FLOPs over the sum of the two durations."""
    rows = ops.gemm_big_rows(ta, tb, M, N, K) if dtype == "f32" else M
    if sample_rows and not ta and ops.gemm_rows_supported(M // sample_rows, sample_rows, N, K):
        rows = M                     # one launch of the per-sample-tile kernel (csrc/gemm_f32_sample.hip)
    parts = [M] if rows in (0, M) else [rows, M - rows]
    n, ms, per = None, 0.0, []
    for m in parts:
        n_i, ms_i = ops.prof_shape(kernel_id, m, N, K)
        if not n_i or (n is not None and n_i != n):
            return None
        n, ms = n_i, ms + ms_i
        per.append({"rows": m, "avg_launch_ms": round(ms_i / n_i, 4)})
    flops = 2.0 * M * N * K
    ach = flops / (ms / n * 1e-3) / 1e12
    traffic, note, source = pmc_lookup(dtype, M, N, K)
    out = {"bound": "mfma", "kernel": what, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
           "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": source, "traffic_note": note,
           "avg_launch_ms": round(ms / n, 4), "launches": n, "flops_per_launch": flops}
    extra = _PMC_EXTRA.get((dtype, M, N, K))
    if extra:
        # counters of the committed --pmc pass of this launch: the matrix pipes' busy fraction and the clock the chip SUSTAINED
        # under it (GRBM_GUI_ACTIVE / time); `peak` above is the boost-clock figure (2.4 GHz), so frac <= clock / 2.4
        out["pmc"] = dict(extra, source=source)
        if "clock_ghz" in extra and dtype == "f32":
            out["pmc"]["peak_at_sustained_clock"] = round(peak * extra["clock_ghz"] / 2.4, 1)
    if len(per) == 2:
        per[0]["kernel"], per[1]["kernel"] = "gemm_f32_big.hip (256x256 tiles, whole rounds of the CUs)", "gemm_f32.hip (128x128 tiles)"
        out["launch_split"] = per
        out["avg_launch_ms_note"] = "sum of the two launches that make up this product"
    return out


class Workload:
    """One BASELINE configuration on one GPU: model, HIP criterion + Adam (solver.py:25-29), synthetic batch."""

    def __init__(self, vqa_amd, model_name, dtype, B, rank, dev, args=None):
        ops = vqa_amd.ops
        self.name, self.dtype, self.B = model_name, dtype, B
        if model_name == "hieCoAtten":
            model = vqa_amd.HieCoAtten(block_num=196, word_num=14, img_size=2048, vocab_size=1000,
                                       embed_size=512, output_size=1000)
        else:
            model = (vqa_amd.MFB if model_name == "mfb" else vqa_amd.MHBCoAtt)(full_cfg(model_name))
        init_like_reference(model)
        model = model.to(dev).train()
        if model_name != "hieCoAtten":
            model.gemm_dtype = {"f32": "fp32", "bf16": "bf16", "bf16-all": "bf16-all"}[dtype]
        if args is not None and args.miopen_lstm and hasattr(model, "use_hip_lstm"):
            model.use_hip_lstm = False
        if args is not None and args.pruned and hasattr(model, "pruned"):
            model.pruned = True
        self.forward_only = bool(args is not None and args.forward_only)
        if self.forward_only:
            model.eval()
        no_overlap = bool(args is not None and args.no_overlap)
        overlap = bool(args is not None and args.overlap)
        self.stream_mode = "one compute stream (projection + fusion one node)" if no_overlap else (
            "two streams (projection on a side stream)" if overlap else
            "one compute stream (projection its own node, weight gradient last)")
        if hasattr(model, "overlap_streams"):              # the SAME configuration at every N
            model.overlap_streams = False if no_overlap else (True if overlap else "same-stream")
            if args is not None and overlap:
                model.side_bf16 = bool(args.side_bf16)
                model.side_cu_limit = int(args.side_cu_limit)
                if args.side_bf16 or args.side_cu_limit:
                    self.stream_mode += " (bf16 projection on the side stream: %s; its GEMMs on <= %s CUs)" % (
                        bool(args.side_bf16), args.side_cu_limit or "all")
        self.model = model
        # solver.py:25-29: criterion + Adam, both on the HIP path (host/train_step.py)
        self.opt = vqa_amd.Adam(model.parameters(), lr=7e-4)
        self.criterion = vqa_amd.train_step.criterion_for(model_name)
        self.img, self.q, self.a = synth_batch(B, rank, dev)
        if dtype != "f32" and model_name != "hieCoAtten":
            # the image grid is stored in bf16 (vqf_cast_f32_bf16 == what
            # FeatureStager(bf16=True) delivers); products accumulate in fp32
            self.img = ops.cast_bf16(self.img.view(-1, self.img.shape[-1])).view(self.img.shape)
        self.soft = torch.softmax(torch.randn((B, 1000), generator=torch.Generator().manual_seed(1236 + rank)), 1).to(dev)
        self.reducer = None

    def step(self):
        m = self.model
        if self.forward_only:
            with torch.no_grad():
                out = m.forward(self.img, self.q)
                return (out[0] if self.name == "hieCoAtten" else out).sum()
        self.opt.zero_grad(set_to_none=True)
        out = m.forward(self.img, self.q)
        if self.name == "hieCoAtten":
            out = out[0]
        loss = self.criterion(out, self.soft if self.name == "mhb_coAtt" else self.a)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        return loss

    def free(self):
        if self.reducer is not None:
            self.reducer.close()
        self.model = self.opt = self.criterion = self.img = self.q = self.a = self.soft = self.reducer = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()


def dominant_mnk(model_name, B):
    """M * N * K of the dominant GEMM of a configuration (image projection; img_emb for HieCoAtten)."""
    return B * 196 * (512 if model_name == "hieCoAtten" else 5000) * 2048


def timed_steps(wl, ops, warmup, steps, fence):
    """W untimed warm-ups, then EXACTLY `steps` timed steps between fences.  Inside the timed region only the dominant GEMM
    launches (M * N * K >= a quarter of the image projection's: both launches of a row-split product) carry hipEvent brackets: a bracket costs the stream ~6-10 us
    between two kernels (rocprofv3 kernel trace of r03: 10.4 us gaps between bracketed launches, none between unbracketed
    ones; ~135 launches per MFB step), so bracketing every kernel would tax the step it measures by ~2 % (13 % for
    HieCoAtten).  The per-kernel table comes from census_steps() afterwards."""
    for _ in range(warmup):
        wl.step()
    ops.prof_reset()
    ops.prof_enable(True, min_mnk=dominant_mnk(wl.name, wl.B) // 4)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = wl.step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.prof_enable(False)
    return elapsed, loss


def exposed_allreduce_steps(wl, steps, fence):
    """`steps` further UNTIMED steps with the reducer's event marks on (one timed hipEvent per bucket on the compute stream in
    finish(): ~10 us each, which is why the timed region runs without them) -> per-bucket exposed all-reduce time, ms."""
    if wl.reducer is None or not getattr(wl.reducer, "active", False):
        return []
    wl.reducer.timing = True
    for _ in range(steps):
        wl.step()
    fence()
    out = wl.reducer.exposed_ms()
    wl.reducer.timing = False
    return out


def census_steps(wl, ops, steps, fence):
    """`steps` further UNTIMED steps with every library kernel bracketed -> {kernel: (launches, total_ms)}."""
    if wl.reducer is not None:
        wl.reducer.timing = False
    ops.prof_reset()
    ops.prof_enable(True)
    for _ in range(steps):
        wl.step()
    fence()
    ops.prof_enable(False)
    rep = ops.prof_report()
    ops.prof_reset()
    return rep


def kernel_table(rep, steps):
    return {k: {"launches_per_step": round(n / steps, 2), "ms_per_step": round(ms / steps, 4)}
            for k, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])}


def step_roofline(model_name, B, ms_per_step, dtype):
    """Whole-step view: algorithmic GEMM FLOPs of the step / MFMA peak of the dtype the large GEMMs run in."""
    tflop = STEP_MFLOP_PER_QA[model_name] * B / 1e6
    peak = BF16_MFMA_PEAK_TFLOPS if dtype != "f32" else FP32_MFMA_PEAK_TFLOPS
    floor_ms = tflop / peak * 1e3
    return {"algorithmic_tflop_per_step": round(tflop, 4), "peak_tflops": peak, "floor_ms": round(floor_ms, 3),
            "frac": round(floor_ms / ms_per_step, 4),
            "note": "floor = all GEMM FLOPs of the step at the MFMA peak of the dtype of its two large GEMM families"
                    + ("; 4 % of the FLOPs (small projections, LSTM) and every HBM-bound stage run in fp32 in this mode, "
                       "and the 512-step batch-axis LSTM recursion is latency-bound: the floor is not attainable"
                       if dtype != "f32" else "")}


def secondary_config(vqa_amd, which, dev, steps, warmup):
    """BASELINE config 3 / 4, timed in this process after the headline."""
    ops = vqa_amd.ops
    if which == "config3":
        name, dtype, B = "mhb_coAtt", "bf16", 512
    elif which == "config3_all":
        name, dtype, B = "mhb_coAtt", "bf16-all", 512
    else:
        name, dtype, B = "hieCoAtten", "f32", 256
    wl = Workload(vqa_amd, name, dtype, B, 0, dev)
    if name == "mhb_coAtt":
        # The 512-step batch-axis LSTM recursion (4.7 ms of latency-bound 4-us kernels) leaves the chip idle: the image
        # projection and its weight gradient run beside it on a second stream, their persistent GEMMs confined to 128
        # CUs so that the recursion's workgroups find free CUs (tools/c3_side_sweep*.sh: 15.8 -> 14.0 ms; 144 / 112 CUs 14.7 / 14.2)
        wl.model.overlap_streams, wl.model.side_bf16, wl.model.side_cu_limit = True, True, C3_SIDE_CUS
        wl.stream_mode = ("two streams: image projection + its weight gradient beside the LSTM recursion, persistent GEMMs on "
                          "<= %d CUs" % C3_SIDE_CUS)

    def fence():
        torch.cuda.synchronize()
    elapsed, loss = timed_steps(wl, ops, warmup, steps, fence)
    ms = 1e3 * elapsed / steps
    if name == "mhb_coAtt":
        M, N, K = B * 196, 5000, 2048
        roof = gemm_roofline(ops, "gemm_bf16", "bf16", M, N, K,
                             "img_conv1d forward GEMM, bf16 operands / fp32 accumulate / bf16 output (M=%d,N=%d,K=%d; "
                             "gemm_bf16_big.hip, 256x256 tiles, LDS-DMA ping-pong, v_mfma_f32_16x16x32_bf16; profiler id "
                             "gemm_bf16)" % (M, N, K), BF16_MFMA_PEAK_TFLOPS)
        if roof is not None:
            w = gemm_roofline(ops, "gemm_bf16", "bf16", N, K, M, "its weight gradient (M=%d,N=%d,K=%d; 32x32x16, K-major "
                              "operands, split-K)" % (N, K, M), BF16_MFMA_PEAK_TFLOPS)
            if w is not None:
                roof["wgrad"] = {k: w[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches", "traffic", "traffic_source")}
        if roof is not None:
            roof["note"] = ("this launch shares the chip with the LSTM recursion (second stream, <= %d of 256 CUs): its duration is "
                            "the price of the overlap, not the kernel's stand-alone rate (1.68 ms = 0.49 of peak on all CUs)" % C3_SIDE_CUS)
        workload = ("MHBCoAtt train step (fwd+KLDiv+bwd+Adam), batch 512, 196x2048 image grid stored in bf16, 14 tokens, "
                    "bf16 operands / fp32 accumulate in %s, everything else fp32; reference LSTM orientation (512-step batch-axis "
                    "recursion)" % ("the img_conv1d and co_att_conv1 GEMM families" if dtype == "bf16" else
                                    "every projection GEMM (img_conv1d, co_att_conv1, ques_proj*, img_proj*, question attention, LSTM input projection)"))
    else:
        M, N, K = B * 196, 512, 2048
        roof = gemm_roofline(ops, "gemm_f32_a0b0(fwd)", "f32", M, N, K,
                             "img_emb forward GEMM + bias + ReLU (M=%d,N=%d,K=%d; hieCoAtten.py:25; gemm_f32_sample.hip: one workgroup "
                             "per sample x 256 columns, 6 row tiles of 32x32x2 + 4 rows on 4x4x1; profiler id gemm_f32_a0b0)" % (M, N, K),
                             FP32_MFMA_PEAK_TFLOPS, sample_rows=196)
        if roof is not None:
            w = gemm_roofline(ops, "gemm_f32_a1b1(wgrad)", "f32", N, K, M, "its weight gradient (M=%d,N=%d,K=%d)" % (N, K, M),
                              FP32_MFMA_PEAK_TFLOPS, ta=1, tb=1)
            if w is not None:
                roof["wgrad"] = {k: w[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches", "traffic", "traffic_source")}
        workload = ("HieCoAtten train step (fwd+CE+bwd+Adam), batch 256, img_size 2048, embed 512, 14 tokens, fp32, "
                    "functional dropout always on (reference behaviour)")
    rep = census_steps(wl, ops, CENSUS_STEPS, fence)
    if name == "mhb_coAtt" and roof is not None:
        roof["standalone"] = standalone_bf16_projection(ops, B, dev)
    out = {"metric": "QA-pairs/sec fwd+bwd, %s batch %d" % (name, B), "value": round(B * steps / elapsed, 2),
           "unit": "QA-pairs/s", "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
           "dtype": "bf16" if dtype != "f32" else "f32",
           "config": {"workload": workload, "global_batch": B, "streams": wl.stream_mode},
           "loss": round(float(loss.item()), 5), "roofline": roof, "step_roofline": step_roofline(name, B, ms, dtype),
           "kernels_ms_per_step": kernel_table(rep, CENSUS_STEPS), "kernels_note": KERNELS_NOTE}
    wl.free()
    return out


def dp_one_rank_child(steps, warmup, pg_timeout=None, timeout_s=600.0):
    """The headline step with the whole data-parallel machinery of host/parallel.py in a ONE-rank RCCL group (solver.py:34-36's
    nn.DataParallel replaced): gradient hooks, bucket copies, asynchronous ncclAllReduce(AVG) on RCCL's high-priority stream,
    large-tile GEMMs launched one workgroup per tile instead of persistent.  What a rank pays for data parallelism before any
    byte crosses xGMI, timed in the driver's own run: the expected N-GPU step is this `ms_per_step` plus the exposed tail of the
    real all-reduce.
    Runs `bench.py --one-rank-group` as a CHILD process, started and awaited BEFORE this process has touched the GPU (fresh
    process: round 5 first timed it in-process behind the three secondary configurations and read 46.5 ms where the stand-alone
    run of the same build gives 40.8 -- gpurun_out/r05/b_one_rank.json; the child's number is the stand-alone one)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--one-rank-group", "--no-cpu-baseline", "--no-secondary", "--steps", str(steps),
           "--warmup", str(warmup)]
    if pg_timeout:
        cmd += ["--pg-timeout", str(pg_timeout)]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "the one-rank child did not finish within %.0f s" % timeout_s}
    lines = [l for l in p.stdout.splitlines() if l.lstrip().startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        return {"error": "one-rank child rc %d: %s" % (p.returncode, (p.stderr or "")[-300:])}
    d = json.loads(lines[0])
    c = d["config"]
    return {"what": "MFB-baseline headline step (B=512, fp32, faithful) under GradientAllReducer in a one-rank RCCL group (hooks, bucket "
                    "copies, %d collectives per step, per-tile GEMM launches), a fresh child process run before the headline" % len(c["allreduce_bucket_bytes"]),
            "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"],
            "gemm_workgroups": c["gemm_workgroups_by_family"]["f32"], "backend": c["backend"],
            "allreduce_bucket_bytes": c["allreduce_bucket_bytes"], "allreduce_exposed_ms": c["allreduce_exposed_ms"],
            "roofline_frac": (d.get("roofline") or {}).get("frac"), "loss": d["loss"]}


def _short(x, n=160):
    return x if not isinstance(x, str) or len(x) <= n else x[:n - 3] + "..."


def compact_line(out, args):
    """The ONE stdout line, kept under ~6 KB (the driver's record keeps the parsed headline keys and the last 2 KB of the
    output): the per-kernel census tables, the secondary configs' full objects and every long note go to
    gpurun_out/bench_census.json (`census_file`); the line ends with `secondary_summary` so that configs 3 and 4 sit in
    the tail the driver stores."""
    full = json.loads(json.dumps(out))
    census_path = os.path.join(ROOT, "gpurun_out", "bench_census.json")
    try:
        os.makedirs(os.path.dirname(census_path), exist_ok=True)
        with open(census_path, "w") as f:
            json.dump(full, f, indent=1)
        census_file = os.path.relpath(census_path, ROOT)
    except OSError as e:
        census_file = "not written: %s" % e
    line = {k: v for k, v in out.items() if k not in ("kernels_ms_per_step", "kernels_note", "secondary", "cpu_baseline",
                                                       "speedup_vs_cpu", "roofline_hbm_kernels")}
    r = line.get("roofline")
    if r:
        r = dict(r)
        for k in ("traffic_note", "avg_launch_ms_note"):
            r.pop(k, None)
        r["kernel"] = _short(r.get("kernel"), 200)
        for sub in ("wgrad", "wgrad_live"):
            if isinstance(r.get(sub), dict):
                r[sub] = {k: _short(v, 120) for k, v in r[sub].items() if k not in ("traffic_note",)}
        line["roofline"] = r
    if line.get("step_roofline"):
        line["step_roofline"] = {k: v for k, v in line["step_roofline"].items() if k != "note"}
    if out.get("hbm_yardsticks"):
        line["hbm_yardsticks"] = dict({k: {"achieved": v["achieved"], "frac": v["frac"]} for k, v in out["hbm_yardsticks"].items()},
                                      _units="GB/s read + written by the library's plain 16-B/lane copy / read-sweep kernels over 2 GB; frac of %d" % HBM_PEAK_GBS)
    hb = out.get("roofline_hbm_kernels") or {}
    line["roofline_hbm_kernels"] = {k: {kk: v[kk] for kk in ("achieved", "frac", "ms_per_step", "traffic", "traffic_over_algorithmic")
                                        if kk in v} for k, v in hb.items()}
    if hb:
        line["roofline_hbm_kernels"]["_units"] = "achieved GB/s of algorithmic bytes; frac of %d GB/s; traffic = counter bytes per step" % HBM_PEAK_GBS
    kern = out.get("kernels_ms_per_step") or {}
    line["launches_per_step"] = round(sum(v["launches_per_step"] for v in kern.values()), 1) if kern else None
    line["top_kernels_ms_per_step"] = {k: v["ms_per_step"] for k, v in list(kern.items())[:8]}
    line["census_file"] = census_file
    if "cpu_baseline" in out:
        c = dict(out["cpu_baseline"])
        c["sample"] = _short(c["sample"], 260)
        for sub in ("with_mask_generation", "config1"):
            if isinstance(c.get(sub), dict):
                c[sub] = {"value": c[sub]["value"], "unit": c[sub]["unit"], "sample": _short(c[sub]["sample"], 110)}
        line["cpu_baseline"] = c
        line["speedup_vs_cpu"] = out.get("speedup_vs_cpu")
    if "secondary" in out:
        summ = {}
        for which, s2 in out["secondary"].items():
            if "error" in s2:
                summ[which] = {"error": _short(s2["error"], 200)}
                continue
            if which == "dp_one_rank":
                summ[which] = {k: s2[k] for k in ("ms_per_step", "plain_ms_per_step", "overhead_frac", "gemm_workgroups",
                                                  "allreduce_exposed_ms", "backend", "steps")}
                continue
            r2 = s2.get("roofline") or {}
            k2 = s2.get("kernels_ms_per_step") or {}
            summ[which] = {"workload": _short(s2["config"]["workload"], 90), "ms_per_step": s2["ms_per_step"], "value": s2["value"],
                           "unit": s2["unit"], "dtype": s2["dtype"], "steps": s2["steps"],
                           "roofline_frac": r2.get("frac"), "roofline_achieved": r2.get("achieved"), "roofline_peak": r2.get("peak"),
                           "roofline_avg_launch_ms": r2.get("avg_launch_ms"), "wgrad_frac": (r2.get("wgrad") or {}).get("frac"),
                           "standalone_frac": (r2.get("standalone") or {}).get("frac"),
                           "step_roofline_frac": (s2.get("step_roofline") or {}).get("frac"),
                           "launches_per_step": round(sum(v["launches_per_step"] for v in k2.values()), 1)}
            summ[which] = {k: v for k, v in summ[which].items() if v is not None}
        line["secondary_summary"] = summ          # LAST: the driver's record keeps the tail of the output
    return line


RANK_LOG_DIR = os.path.join(ROOT, "gpurun_out")
READY_MARK = "process group ready"


def stage(msg):
    """Progress marker of a rank: one line on stderr (the per-rank log under spawn_ranks) and, under any launcher, the last
    marker in gpurun_out/rank<r>.stage -- what a peer's rendezvous time-out and the parent's deadline message quote."""
    r = os.environ.get("RANK", "0")
    line = "[bench rank %s +%.1fs] %s" % (r, time.time() - _T0, msg)
    sys.stderr.write(line + "\n")
    sys.stderr.flush()
    if "WORLD_SIZE" in os.environ:
        try:
            os.makedirs(RANK_LOG_DIR, exist_ok=True)
            with open(os.path.join(RANK_LOG_DIR, "rank%s.stage" % r), "w") as f:
                f.write(line + "\n")
        except OSError:
            pass


def peer_stages(world):
    """{rank: last stage marker} as the ranks of this launch left them (a rank that never started has none)."""
    out = {}
    for r in range(world):
        try:
            st = os.stat(os.path.join(RANK_LOG_DIR, "rank%d.stage" % r))
            txt = open(os.path.join(RANK_LOG_DIR, "rank%d.stage" % r)).read().strip()
            out[r] = txt if st.st_mtime >= _T0 - 3600 else "(stale marker) " + txt
        except OSError:
            out[r] = "no marker: the rank never reached bench.py's main()"
    return out


def _tail(path, n=40):
    try:
        with open(path, errors="replace") as f:
            return f.read().splitlines()[-n:]
    except OSError as e:
        return ["(no log: %s)" % e]


def spawn_ranks(n, launch_timeout):
    """`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment): this process -- which has not touched
    the GPU and never will -- starts N fresh ranks as CHILDREN (no exec; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torchrun
    sets them), each in its own session with stdout + stderr in gpurun_out/rank<r>.log, and watches them:
      * every rank must log "process group ready" within `launch_timeout` seconds of the start (the first `import torch` on a
        fresh box takes 1-2 minutes; the rendezvous itself is bounded by host/parallel.py's 120 s);
      * afterwards the job may not go `launch_timeout` seconds without ANY rank's log growing or a rank ending;
      * a rank that ends non-zero ends the job: its peers get 15 s to follow, then their process groups are killed.
    On any of these the parent terminates every child process group, says which ranks had not reached which stage, prints the
    last 40 lines of each rank's log and exits non-zero (124 for a deadline) -- a stalled rank on an 8-GPU lease ends with a
    diagnosis, not with a silent time-limit kill.  On success rank 0's ONE JSON line is relayed on stdout."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.makedirs(RANK_LOG_DIR, exist_ok=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               PYTHONUNBUFFERED="1")
    logs = [os.path.join(RANK_LOG_DIR, "rank%d.log" % r) for r in range(n)]
    for r in range(n):
        try:
            os.remove(os.path.join(RANK_LOG_DIR, "rank%d.stage" % r))
        except OSError:
            pass
    procs = []
    for r in range(n):
        f = open(logs[r], "w")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], cwd=ROOT,
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0", ROLE_RANK=str(r)),
                                      stdout=f, stderr=subprocess.STDOUT, start_new_session=True))
        f.close()

    def ready(r):
        try:
            return READY_MARK in open(logs[r], errors="replace").read()
        except OSError:
            return False

    def sizes():
        out = []
        for lg in logs:
            try:
                out.append(os.path.getsize(lg))
            except OSError:
                out.append(0)
        return out

    def kill_all():
        for sig, wait in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            alive = [p for p in procs if p.poll() is None]
            if not alive:
                return
            for p in alive:
                try:
                    os.killpg(p.pid, sig)
                except OSError:
                    pass
            t_end = time.time() + wait
            while time.time() < t_end and any(p.poll() is None for p in alive):
                time.sleep(0.1)

    def report(reason):
        sys.stderr.write("bench.py: %s\n" % reason)
        st = peer_stages(n)
        for r in range(n):
            rc = procs[r].poll()
            sys.stderr.write("  rank %d: %s; exit code %s; last marker: %s\n"
                             % (r, "reached '%s'" % READY_MARK if ready(r) else "NEVER reached '%s'" % READY_MARK,
                                "none (killed by the parent)" if rc is None else rc, st[r]))
        for r in range(n):
            sys.stderr.write("----- last 40 lines of %s -----\n" % os.path.relpath(logs[r], ROOT))
            for line in _tail(logs[r]):
                sys.stderr.write("  " + line + "\n")
        sys.stderr.flush()

    t0 = time.time()
    last_growth, last_sizes, last_beat = t0, sizes(), t0
    all_ready = False
    rc = None
    while True:
        time.sleep(0.25)
        now = time.time()
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            rc = max((abs(c) for c in codes), default=0)
            if rc:
                report("rank(s) %s ended non-zero" % [r for r, c in enumerate(codes) if c])
            break
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            t_end = now + 15.0
            while time.time() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.25)
            report("rank(s) %s ended non-zero (exit code %s) while rank(s) %s were still running"
                   % (bad, [codes[r] for r in bad], [r for r, p in enumerate(procs) if p.poll() is None]))
            kill_all()
            rc = abs(codes[bad[0]]) or 1
            break
        sz = sizes()
        if sz != last_sizes:
            last_sizes, last_growth = sz, now
        if not all_ready:
            all_ready = all(ready(r) for r in range(n))
            if not all_ready and now - t0 > launch_timeout:
                st = peer_stages(n)
                late = [r for r in range(n) if not ready(r)]
                waiting = [r for r in late if "] rendezvous:" in st[r]]         # arrived, blocked on their peers
                absent = [r for r in late if r not in waiting]
                report("launch deadline: rank(s) %s had not reached '%s' %.0f s after the start (--launch-timeout %.0f); "
                       "rank(s) %s never ARRIVED at the rendezvous, rank(s) %s were waiting in it for their peers"
                       % (late, READY_MARK, now - t0, launch_timeout, absent, waiting))
                kill_all()
                rc = 124
                break
        elif now - last_growth > launch_timeout:
            report("stall: no rank has written to its log or ended for %.0f s (--launch-timeout %.0f)"
                   % (now - last_growth, launch_timeout))
            kill_all()
            rc = 124
            break
        if now - last_beat > 60.0:
            last_beat = now
            st = peer_stages(n)
            sys.stderr.write("bench.py: %d ranks running for %.0f s; %s\n"
                             % (n, now - t0, "; ".join("rank %d: %s" % (r, st[r].split("] ", 1)[-1]) for r in range(n))))
            sys.stderr.flush()
    lines = [l for l in _tail(logs[0], 100000) if l.lstrip().startswith("{")]
    if rc == 0:
        for line in lines:
            print(line, flush=True)
        if len(lines) != 1:
            report("expected one JSON line from rank 0, got %d" % len(lines))
            rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="QA pairs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=512, help="batch of the cpu_baseline train step (default: the metric's 512)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the in-process runs of BASELINE configs 3 and 4 that follow the headline on a default 1-GPU run")
    ap.add_argument("--secondary-steps", type=int, default=12)
    ap.add_argument("--secondary-warmup", type=int, default=4)
    ap.add_argument("--no-overlap", action="store_true", help="A/B: projection + fusion as ONE autograd node")
    ap.add_argument("--overlap", action="store_true",
                    help="A/B: image projection on a side HIP stream.  Default at EVERY N: one compute stream with "
                         "the projection as its own autograd node ('same-stream'): its weight-gradient GEMM runs "
                         "last in the backward and the other buckets' all-reduce (RCCL's stream) hides behind it")
    ap.add_argument("--no-defer", action="store_true",
                    help="A/B: one-stream form with the projection's product issued FIRST in the forward (round 2) instead of behind the question encoder")
    ap.add_argument("--side-bf16", action="store_true", help="with --overlap: the bf16 projection goes to the side stream too (MFB.side_bf16)")
    ap.add_argument("--side-cu-limit", type=int, default=0, help="with --overlap: the side stream's persistent GEMMs use at most this many CUs")
    ap.add_argument("--miopen-lstm", action="store_true", help="A/B: question-encoder LSTM on nn.LSTM (MIOpen)")
    ap.add_argument("--pruned", action="store_true",
                    help="MFB.pruned: skip the work that is provably dead under the reference's singleton-axis softmaxes "
                         "(bit-identical results).  NOT the headline: the default executes everything the reference does")
    ap.add_argument("--forward-only", action="store_true",
                    help="BASELINE config 1 shape of work: forward pass only (eval mode, no_grad); not the headline")
    ap.add_argument("--model", default="mfb", choices=["mfb", "mhb_coAtt", "hieCoAtten"],
                    help="mfb = the headline (BASELINE config 2/5); the others time configs 3 and 4")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "bf16-all"],
                    help="bf16 = bf16 operands in the two large GEMM families (BASELINE config 3 mode); bf16-all = in every "
                         "projection GEMM (ques_proj*, img_proj*, question attention too)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); "
                    "'gloo' lets several ranks share one GPU for rehearsals")
    ap.add_argument("--gemm-workgroups", default=None, choices=["per-tile", "persistent"],
                    help="large-tile GEMM launch form under data parallelism (default: per-tile at N > 1, see host/parallel.py)")
    ap.add_argument("--one-rank-group", action="store_true",
                    help="N = 1 only: run the data-parallel machinery anyway (a one-rank RCCL group, buckets, hooks, per-tile "
                         "GEMM launches): what data parallelism costs a rank before any byte crosses xGMI")
    ap.add_argument("--launch-timeout", type=float, default=420.0,
                    help="bare `--gpus N` launch: seconds every rank has to reach 'process group ready', and afterwards the longest "
                         "silence (no rank log growing, no rank ending) before the parent kills the job with a diagnosis")
    ap.add_argument("--pg-timeout", type=float, default=None,
                    help="bound in seconds on the torch.distributed rendezvous and on each collective (default 120: host/parallel.py)")
    ap.add_argument("--final-two-streams", action="store_true",
                    help="A/B: the two products of a final MFB block (and their gradients) on two streams (functions.FinalMfbFn.TWO_STREAMS)")
    ap.add_argument("--no-dp-one-rank", action="store_true",
                    help="skip the one-rank data-parallel rehearsal (`secondary_summary.dp_one_rank`) of a default 1-GPU run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, args.launch_timeout))       # before anything here has initialised the GPU

    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
    dp_one = None
    if (not multi and args.gpus == 1 and args.model == "mfb" and args.dtype == "f32" and args.batch == 512 and not args.pruned
            and not args.forward_only and not args.miopen_lstm and not args.no_secondary and not args.one_rank_group
            and not args.no_dp_one_rank):
        # the one-rank data-parallel rehearsal of a default run: a fresh child, finished before this process touches the GPU
        dp_one = dp_one_rank_child(args.secondary_steps, args.secondary_warmup, args.pg_timeout)
    if multi:
        # In-rank watchdog for launches this file does not supervise (the driver's torchrun line): a C-level timer thread that
        # needs no GIL -- it fires inside a blocked rendezvous or RCCL bootstrap too --, dumps every thread's stack to the rank's
        # stderr and ends the process (exit code 1; torchrun then ends the peers).  Armed for the launch phase here and again
        # for the timed loop; the last marker in gpurun_out/rank<r>.stage says how far the rank got.
        import faulthandler
        faulthandler.dump_traceback_later(args.launch_timeout, exit=True)
        stage("start, pid %d; watchdog %.0f s for the launch phase" % (os.getpid(), args.launch_timeout))
    import vqa_amd
    from importlib import import_module
    parallel = import_module("vqa-attention-networks_amd.host.parallel")
    ops = vqa_amd.ops
    vqa_amd.lib.load()
    if multi:
        stage("torch imported, libvqa_fusion.so loaded")

    if args.no_defer:
        import_module("vqa-attention-networks_amd.host.mfb")._SideStream.DEFER = False
    if args.final_two_streams:
        vqa_amd.functions.FinalMfbFn.TWO_STREAMS = True
    if os.environ.get("VQF_TEST_STALL_RANK") == os.environ.get("RANK", "0") and multi:
        # test switch (tests/test_launch_deadline.py): this rank never arrives at the rendezvous
        stage("VQF_TEST_STALL_RANK: sleeping %s s before the rendezvous" % os.environ.get("VQF_TEST_STALL_S", "600"))
        time.sleep(float(os.environ.get("VQF_TEST_STALL_S", "600")))
    if multi:
        stage("rendezvous: init_process_group(%s), bound %s s" % (args.backend or "nccl", args.pg_timeout or parallel.PG_TIMEOUT_S))
    try:
        rank, world, local = parallel.init_distributed(args.backend, force=args.one_rank_group, timeout_s=args.pg_timeout)
    except RuntimeError as e:
        if multi:       # name the ranks that never got to the rendezvous: their last markers (gpurun_out/rank<r>.stage)
            st = peer_stages(int(os.environ["WORLD_SIZE"]))
            sys.stderr.write("bench.py: %s\n" % e)
            for r, m in st.items():
                sys.stderr.write("  rank %d last marker: %s\n" % (r, m))
            sys.stderr.flush()
            sys.exit(3)
        raise
    if multi:
        stage("%s (world %d, backend %s)" % (READY_MARK, world, dist.get_backend()))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: an explicit WORLD_SIZE must equal --gpus (unset it and bench.py starts "
                         "the N ranks itself, or launch them with `python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...`)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B = args.batch
    if args.model == "hieCoAtten" and args.batch == 512:
        B = 256                                       # BASELINE config 4
    wl = Workload(vqa_amd, args.model, args.dtype, B, rank, dev, args)
    reducer = parallel.GradientAllReducer(wl.model, gemm_workgroups=args.gemm_workgroups,   # broadcasts rank 0's weights; no-op at world 1
                                          single_rank=args.one_rank_group)
    wl.reducer = reducer
    if multi:
        faulthandler.cancel_dump_traceback_later()
        loop_bound = args.launch_timeout + 2.0 * (args.warmup + args.steps + 2 * CENSUS_STEPS)
        faulthandler.dump_traceback_later(loop_bound, exit=True)
        stage("parameters broadcast, %d gradient buckets; watchdog %.0f s for the step loops" % (len(reducer.buckets), loop_bound))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, loss = timed_steps(wl, ops, args.warmup, args.steps, fence)
    if multi:
        stage("%d warm-up + %d timed steps done (%.1f ms per step on this rank)" % (args.warmup, args.steps, 1e3 * elapsed / args.steps))
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = B * world * args.steps / elapsed

    # ---- roofline of the dominant kernel: the image-projection GEMM (mfb.py:96 / mhb_coAtt.py:98) or, for HieCoAtten,
    # the img_emb GEMM (hieCoAtten.py:25); fp32 or bf16 launch, whichever this run executed ----------------
    is_bf16 = args.dtype != "f32" and args.model != "hieCoAtten"
    Nn = 512 if args.model == "hieCoAtten" else 5000
    M, N, K = B * 196, Nn, 2048
    sample_rows = 196 if args.model == "hieCoAtten" else 0
    peak = BF16_MFMA_PEAK_TFLOPS if is_bf16 else FP32_MFMA_PEAK_TFLOPS
    fam = ("gemm_bf16", "gemm_bf16") if is_bf16 else ("gemm_f32_a0b0(fwd)", "gemm_f32_a1b1(wgrad)")
    proj = "img_emb" if args.model == "hieCoAtten" else "img_conv1d"
    what = ("%s forward GEMM (M=%d,N=%d,K=%d; gemm_bf16_big.hip, 256x256 tiles, LDS-DMA ping-pong, bf16 operands / fp32 "
            "accumulate; profiler id gemm_bf16)" % (proj, M, N, K)) if is_bf16 else (
        "%s forward GEMM (M=%d,N=%d,K=%d; %s; profiler id gemm_f32_a0b0)"
        % (proj, M, N, K, "gemm_f32_big.hip, 256x256 tiles, LDS-DMA, staggered wave halves" if Nn == 5000
           else "gemm_f32_sample.hip: one workgroup per sample x 256 columns"))
    roofline = gemm_roofline(ops, fam[0], "bf16" if is_bf16 else "f32", M, N, K, what, peak, sample_rows=sample_rows)
    flops = 2.0 * M * N * K
    if roofline is not None:
        w = gemm_roofline(ops, fam[1], "bf16" if is_bf16 else "f32", N, K, M,
                          "%s img_conv1d wgrad (K-major operands, split-K)" % fam[1], peak, ta=1, tb=1)
        if w is not None:
            roofline["wgrad"] = {"kernel": w["kernel"], "achieved": w["achieved"], "frac": w["frac"],
                                 "avg_launch_ms": w["avg_launch_ms"], "launches": w["launches"],
                                 "operands": ("in-step launch; in faithful MFB dP is EXACTLY ZERO (singleton-axis "
                                              "softmax): see wgrad_live for random operands")
                                 if (args.model == "mfb" and not args.pruned) else "in-step launch, live operands"}
    exposed = exposed_allreduce_steps(wl, CENSUS_STEPS, fence)
    rep = census_steps(wl, ops, CENSUS_STEPS, fence)
    gemm_wg = reducer.gemm_workgroups()
    if roofline is not None and world == 1 and args.dtype == "f32" and args.model == "mfb" and not args.forward_only:
        n_l, ms_l, red_l = live_wgrad_probe(ops, B, dev)
        if n_l:
            achl = flops / (ms_l / n_l * 1e-3) / 1e12
            roofline["wgrad_live"] = {"kernel": "the same launch (ops.gemm(dP, X, ta, tb), M=%d N=%d K=%d) on uniform "
                                                "random dP / relu(N(0,1)) image, outside the step" % (N, K, M),
                                      "achieved": round(achl, 2), "frac": round(achl / FP32_MFMA_PEAK_TFLOPS, 4),
                                      "avg_launch_ms": round(ms_l / n_l, 4), "launches": n_l,
                                      "splitk_reduce_ms_per_launch": round(red_l / n_l, 4)}
            tr, _, src = pmc_lookup("f32", N, K, M)       # beyond-L2 bytes of this launch from the committed --pmc passes
            if tr is not None:
                roofline["wgrad_live"].update(traffic=tr, traffic_source=src)
    kernels = kernel_table(rep, CENSUS_STEPS)
    yardstick = None
    if world == 1 and args.dtype == "f32" and args.model == "mfb" and not args.forward_only and B == 512:
        yardstick = hbm_yardsticks(ops, B, dev)
    # secondary roofline: the HBM-bound kernels of the step.  algorithmic bytes per step: fusion fwd reads P (+q) and
    # writes R for the L=196 stage and the final block; bwd reads P, dY, Y and writes dP; the glimpse passes
    # read the image tensor once; att_logits_bwd reads + writes the co-attention hidden layer.  `traffic` = HBM bytes per
    # step from the committed rocprofv3 --pmc passes (profiles/r*_pmc_fuse.json), null until such a pass exists.
    rows, o5 = B * 196, 5000
    pbytes = 2.0 if is_bf16 else 4.0
    hid = 1024 if args.model == "mfb" else 512
    alg = {
        "mfb_fuse_fwd": pbytes * rows * o5 + 4.0 * (B * o5 + rows * 1000 + rows) + 4.0 * (2 * B * o5 + B * 1000),
        "mfb_fuse_bwd": 2 * pbytes * rows * o5 + 4.0 * (2 * rows * 1000 + 2 * B * o5) + 4.0 * (3 * B * o5 + 2 * B * 1000),
        "glimpse_pool_fwd": pbytes * rows * 2048 + 4.0 * (B * 14 * 1024),
        "glimpse_pool_bwd": pbytes * rows * 2048 + 4.0 * (B * 14 * 1024 * 2),
        "att_logits_bwd": 4.0 * 2 * (rows * hid + B * 14 * hid),
        "scale_rows": 4.0 * 2 * rows * 1000, "rowdot": 4.0 * 2 * rows * 1000,
    }
    roofline_hbm = {}
    if args.model != "hieCoAtten":
        for name, nbytes in alg.items():
            if name in rep and rep[name][1] > 0:
                ms_step = rep[name][1] / CENSUS_STEPS
                gbs = nbytes / (ms_step * 1e-3) / 1e9
                if name in ("scale_rows", "rowdot") and gbs > HBM_PEAK_GBS:
                    continue        # the (B*196, 1000) pass is folded into co_att_conv1 (NormLink): what is left is the final block's small launch
                pm, src = pmc_kernel_lookup(name) if (args.model == "mfb" and args.dtype == "f32" and B == 512) else (None, None)
                roofline_hbm[name] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(gbs / HBM_PEAK_GBS, 4), "ms_per_step": round(ms_step, 4),
                                      "algorithmic_bytes_per_step": nbytes,
                                      "traffic": pm["traffic_bytes"] if pm else None, "traffic_source": src,
                                      "traffic_over_algorithmic": round(pm["traffic_bytes"] / nbytes, 3) if pm else None}

    out = None
    if rank == 0:
        out = {
            "metric": ("QA-pairs/sec fwd+bwd, MFB-baseline batch 512" if (args.model == "mfb" and B == 512)
                       else "QA-pairs/sec fwd+bwd, %s batch %d" % (args.model, B)) if not args.forward_only
                      else "QA-pairs/sec forward only, %s batch %d" % (args.model, B),
            "value": round(value, 2), "unit": "QA-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype != "f32" else "f32", "data": "synthetic",
            "config": {"workload": ("forward pass only (eval), " if args.forward_only else "") + (
                                   "MFB-baseline train step (fwd+loss+bwd+grad all-reduce+Adam), "
                                   "batch %d per GPU, 196x2048 image grid, 14 tokens, %s, mode=%s"
                                   % (B, "fp32" if args.dtype == "f32" else ("bf16 operands / fp32 accumulate" + (" in every projection GEMM" if args.dtype == "bf16-all" else "")),
                                      "pruned (NOT the headline: provably dead work skipped)" if args.pruned else "faithful")
                                   if args.model == "mfb" else "%s train step, batch %d per GPU, %s" % (args.model, B, args.dtype)),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "grad_allreduce_bytes": reducer.gradient_bytes(),
                       "ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
                       "backend": (dist.get_backend() if dist.is_initialized() else "none (single process)"),
                       "streams": wl.stream_mode,
                       "gemm_workgroups": ("%s (data-parallel default: CUs free up for the collective)" % gemm_wg["f32"]
                                           if gemm_wg["f32"].startswith("one per tile") else gemm_wg["f32"]),
                       "gemm_workgroups_by_family": gemm_wg,
                       "allreduce_bucket_bytes": reducer.bucket_bytes_list(),
                       "allreduce_exposed_ms": exposed},
            "loss": round(float(loss.item()), 5),
            "roofline": roofline,
            "step_roofline": step_roofline(args.model, B, ms_per_step, args.dtype) if not (args.forward_only or args.pruned) else None,
            "roofline_hbm_kernels": roofline_hbm,
            "hbm_yardsticks": yardstick,
            "kernels_ms_per_step": kernels, "kernels_note": KERNELS_NOTE,
        }
    headline_default = (args.model == "mfb" and args.dtype == "f32" and B == 512 and not args.pruned
                        and not args.forward_only and not args.miopen_lstm)
    if world == 1 and headline_default and not args.no_secondary and not args.one_rank_group:
        wl.free()
        del reducer
        out["secondary"] = {}
        for which in ("config3", "config4", "config3_all"):
            try:
                out["secondary"][which] = secondary_config(vqa_amd, which, dev, args.secondary_steps, args.secondary_warmup)
            except Exception as e:          # the headline line must survive a secondary failure; say what happened
                out["secondary"][which] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if dp_one is not None:
            if "error" not in dp_one:
                dp_one["plain_ms_per_step"] = round(ms_per_step, 3)
                dp_one["overhead_frac"] = round(dp_one["ms_per_step"] / ms_per_step - 1.0, 4)
            out["secondary"]["dp_one_rank"] = dp_one
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.one_rank_group:
            out["cpu_baseline"] = cpu_baseline(batch=args.cpu_batch)
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(compact_line(out, args)), flush=True)
    if multi:
        faulthandler.cancel_dump_traceback_later()
        stage("line printed" if rank == 0 else "done")
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
