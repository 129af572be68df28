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
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak (= vector peak)
HBM_PEAK_GBS = 8000.0


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


def cpu_baseline(batch=512, timed=3, config1_timed=5):
    """The oracle (CPU restatement, PyTorch CPU ops) on the host cores, on the metric's own configuration
    (BASELINE.md section 3): the MFB train step (fwd + loss + bwd + Adam, dropout masks supplied) at B=512,
    1 warm-up + `timed` steps, median; plus BASELINE config 1 (MFB forward, eval, B=32) as `config1`.
    About 45 s on the 16 cores of a GPU box; `--cpu-batch` shrinks the sample for quick runs."""
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
        drop = dict(l=(torch.rand((Bs, 14, 1024), generator=gen) >= 0.3),
                    m1=(torch.rand((Bs, 196, 5000), generator=gen) >= 0.1),
                    m2=(torch.rand((Bs, 5000), generator=gen) >= 0.1))
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        logits = O.mfb_forward(sd, cfg, img, q, drop=drop)
        loss = O.ce_loss(logits, a)
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    def fwd_only(img, q):
        t0 = time.perf_counter()
        with torch.no_grad():
            O.mfb_forward(sd, cfg, img, q)              # eval: no dropout (BASELINE config 1)
        return time.perf_counter() - t0

    img, q, a = synth_batch(batch, 0, "cpu")
    one_step(img[:16], q[:16], a[:16])                      # page-in / thread-pool warm-up
    one_step(img, q, a)                                     # 1 warm-up at the full batch
    times = sorted(one_step(img, q, a) for _ in range(timed))
    t = times[len(times) // 2]
    fwd_only(img[:32], q[:32])
    t1s = sorted(fwd_only(img[:32], q[:32]) for _ in range(config1_timed))
    t1 = t1s[len(t1s) // 2]
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": round(batch / t, 3), "unit": "QA-pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle MFB train step (fwd+loss+bwd+Adam, dropout masks supplied), B=%d, "
                      "1 warm-up + %d timed steps, median %.2f s/step, %d threads; CPU: %s"
                      % (batch, timed, t, cores, model),
            "config1": {"value": round(32 / t1, 3), "unit": "QA-pairs/s",
                        "sample": "BASELINE config 1: oracle MFB forward, eval, B=32, 1 warm-up + %d timed, "
                                  "median %.3f s, %d threads" % (config1_timed, t1, cores)}}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="QA pairs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=512, help="batch of the cpu_baseline train step (default: the metric's 512)")
    ap.add_argument("--no-overlap", action="store_true", help="A/B: projection + fusion as ONE autograd node")
    ap.add_argument("--overlap", action="store_true",
                    help="A/B: image projection on a side HIP stream.  Default at EVERY N: one compute stream with "
                         "the projection as its own autograd node ('same-stream'): its weight-gradient GEMM runs "
                         "last in the backward and the other buckets' all-reduce (RCCL's stream) hides behind it")
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
    args = ap.parse_args()

    import vqa_amd
    from importlib import import_module
    parallel = import_module("vqa-attention-networks_amd.host.parallel")
    ops = vqa_amd.ops
    vqa_amd.lib.load()

    rank, world, local = parallel.init_distributed(args.backend)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N ranks with `python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...`"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B = args.batch
    if args.model == "hieCoAtten":
        if args.batch == 512:
            B = 256                                   # BASELINE config 4
        model = vqa_amd.HieCoAtten(block_num=196, word_num=14, img_size=2048, vocab_size=1000,
                                   embed_size=512, output_size=1000)
    else:
        cfg = full_cfg(args.model)
        model = (vqa_amd.MFB if args.model == "mfb" else vqa_amd.MHBCoAtt)(cfg)
    init_like_reference(model)
    model = model.to(dev).train()
    if args.model != "hieCoAtten":
        model.gemm_dtype = {"f32": "fp32", "bf16": "bf16", "bf16-all": "bf16-all"}[args.dtype]
    if args.miopen_lstm and hasattr(model, "use_hip_lstm"):
        model.use_hip_lstm = False
    if args.pruned and hasattr(model, "pruned"):
        model.pruned = True
    if args.forward_only:
        model.eval()
    stream_mode = "one compute stream (projection + fusion one node)" if args.no_overlap else (
        "two streams (projection on a side stream)" if args.overlap else
        "one compute stream (projection its own node, weight gradient last)")
    if hasattr(model, "overlap_streams"):                # the SAME configuration at every N
        model.overlap_streams = False if args.no_overlap else (True if args.overlap else "same-stream")
    reducer = parallel.GradientAllReducer(model)        # broadcasts rank 0's weights; no-op at world 1
    # solver.py:25-29: criterion + Adam, both on the HIP path (host/train_step.py)
    opt = vqa_amd.Adam(model.parameters(), lr=7e-4)
    criterion = vqa_amd.train_step.criterion_for(args.model)
    img, q, a = synth_batch(B, rank, dev)
    if args.dtype != "f32" and args.model != "hieCoAtten":
        # the image grid is stored in bf16 (vqf_cast_f32_bf16 == what
        # FeatureStager(bf16=True) delivers); products accumulate in fp32
        img = ops.cast_bf16(img.view(-1, img.shape[-1])).view(img.shape)
    soft = torch.softmax(torch.randn((B, 1000), generator=torch.Generator().manual_seed(1236 + rank)), 1).to(dev)

    def fwd_step():
        with torch.no_grad():
            out = model.forward(img, q)
            return (out[0] if args.model == "hieCoAtten" else out).sum()

    def step():
        if args.forward_only:
            return fwd_step()
        opt.zero_grad(set_to_none=True)
        out = model.forward(img, q)
        if args.model == "hieCoAtten":
            out = out[0]
        loss = criterion(out, soft if args.model == "mhb_coAtt" else a)
        loss.backward()
        reducer.finish()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    reducer.timing = True
    ops.prof_reset()
    ops.prof_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.prof_enable(False)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = B * world * args.steps / elapsed

    # ---- roofline of the dominant kernel: the image-projection GEMM (mfb.py:96) --------------
    M, N, K = B * 196, 5000, 2048
    n_f, ms_f = ops.prof_gemm(0, 0, M, N, K)               # forward projection
    n_w, ms_w = ops.prof_gemm(1, 1, N, K, M)               # its weight gradient (same FLOPs)
    flops = 2.0 * M * N * K
    roofline = None
    if n_f:
        ach = flops / (ms_f / n_f * 1e-3) / 1e12
        traffic, traffic_note, traffic_source = None, None, None
        # PMC passes cannot run inside the timed process: the committed rocprofv3 --pmc result of this launch
        for src in ("profiles/r02_pmc_gemm.json", "profiles/r01_pmc_gemm.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, src)))
                if (pmc["M"], pmc["N"], pmc["K"]) == (M, N, K):
                    traffic, traffic_note, traffic_source = pmc["traffic_bytes"], pmc["note"] + "; " + pmc["formula"], src
                    break
            except Exception:
                pass
        roofline = {"bound": "mfma", "kernel": "img_conv1d forward GEMM (M=%d,N=%d,K=%d; gemm_f32_big.hip, 256x256 tiles, LDS-DMA, staggered wave halves; profiler id gemm_f32_a0b0)" % (M, N, K),
                    "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": traffic_source, "traffic_note": traffic_note,
                    "avg_launch_ms": round(ms_f / n_f, 4), "launches": n_f,
                    "flops_per_launch": flops}
        if n_w:
            achw = flops / (ms_w / n_w * 1e-3) / 1e12
            roofline["wgrad"] = {"kernel": "gemm_f32_a1b1 img_conv1d wgrad (gemm_f32_big.hip, K-major interleaved strips, split-K 8)", "achieved": round(achw, 2),
                                 "frac": round(achw / FP32_MFMA_PEAK_TFLOPS, 4),
                                 "avg_launch_ms": round(ms_w / n_w, 4), "launches": n_w,
                                 "operands": ("in-step launch; in faithful MFB dP is EXACTLY ZERO (singleton-axis "
                                              "softmax): see wgrad_live for random operands")
                                 if (args.model == "mfb" and not args.pruned) else "in-step launch, live operands"}
    rep = ops.prof_report()
    exposed = reducer.exposed_ms()
    if roofline is not None and world == 1 and args.dtype == "f32" and not args.forward_only:
        n_l, ms_l, red_l = live_wgrad_probe(ops, B, dev)
        if n_l:
            achl = flops / (ms_l / n_l * 1e-3) / 1e12
            roofline["wgrad_live"] = {"kernel": "the same launch (ops.gemm(dP, X, ta, tb), M=%d N=%d K=%d) on uniform "
                                                "random dP / relu(N(0,1)) image, outside the step" % (N, K, M),
                                      "achieved": round(achl, 2), "frac": round(achl / FP32_MFMA_PEAK_TFLOPS, 4),
                                      "avg_launch_ms": round(ms_l / n_l, 4), "launches": n_l,
                                      "splitk_reduce_ms_per_launch": round(red_l / n_l, 4)}
            try:        # beyond-L2 bytes of this launch from the committed rocprofv3 --pmc passes
                pw = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_wgrad.json")))
                if (pw["M"], pw["N"], pw["K"]) == (N, K, M):
                    roofline["wgrad_live"].update(traffic=pw["traffic_bytes"], traffic_source="profiles/r02_pmc_wgrad.json")
            except Exception:
                pass
    kernels = {k: {"launches_per_step": round(n / args.steps, 2), "ms_per_step": round(ms / args.steps, 4)}
               for k, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])}
    # secondary roofline: the HBM-bound MFB fusion kernels (mfb.py:98-106 and its backward).
    # algorithmic bytes per step: fwd reads P (+q) and writes R for the L=196 stage and the final block;
    # bwd reads P, dY, Y and writes dP.
    rows, o5 = B * 196, 5000
    fwd_bytes = 4.0 * (rows * o5 + B * o5 + rows * 1000 + rows) + 4.0 * (2 * B * o5 + B * 1000)
    bwd_bytes = 4.0 * (2 * rows * o5 + 2 * rows * 1000 + 2 * B * o5) + 4.0 * (3 * B * o5 + 2 * B * 1000)
    roofline_hbm = {}
    for name, nbytes in (("mfb_fuse_fwd", fwd_bytes), ("mfb_fuse_bwd", bwd_bytes)):
        if name in rep and rep[name][1] > 0:
            ms_step = rep[name][1] / args.steps
            gbs = nbytes / (ms_step * 1e-3) / 1e9
            roofline_hbm[name] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(gbs / HBM_PEAK_GBS, 4), "ms_per_step": round(ms_step, 4),
                                  "algorithmic_bytes_per_step": nbytes}

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
                       "streams": stream_mode,
                       "gemm_workgroups": ("one per tile (data-parallel default: CUs free up for the collective)"
                                           if os.environ.get("VQF_GEMM_F32_PERSIST") == "0" else "persistent, one per CU"),
                       "allreduce_bucket_bytes": reducer.bucket_bytes_list(),
                       "allreduce_exposed_ms": exposed},
            "loss": round(float(loss.item()), 5),
            "roofline": roofline,
            "roofline_hbm_kernels": roofline_hbm,
            "kernels_ms_per_step": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch=args.cpu_batch)
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
