#!/usr/bin/env python3
"""Which torch (at::native) device kernels are still launched inside a train step of the path, and from which line.
Runs a few warm steps of one bench workload, then profiles ONE step with torch.profiler (CPU side, with stacks) and
lists every aten op that ran on a cuda tensor together with the innermost repo frame that issued it.
    python tools/native_ops.py --model mfb|mhb_coAtt|hieCoAtten [--dtype f32|bf16] [--batch 512]"""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="mfb")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()
import importlib  # noqa: E402
vqa_amd = importlib.import_module("vqa_amd")
dev = torch.device("cuda:0")
wl = bench.Workload(vqa_amd, args.model, args.dtype, args.batch, 0, dev)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    wl.step()
    torch.cuda.synchronize()
# leaf aten ops only (an op whose time range holds no other aten op) -- those are the ones that launch a kernel
evs = [e for e in prof.events() if e.name.startswith("aten::")]
evs.sort(key=lambda e: (e.time_range.start, -e.time_range.end))
leaves = []
for i, e in enumerate(evs):
    nxt = evs[i + 1] if i + 1 < len(evs) else None
    if nxt is not None and nxt.time_range.start < e.time_range.end and nxt.thread == e.thread:
        continue
    leaves.append(e)
SKIP = {"aten::empty", "aten::empty_like", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::_unsafe_view",
        "aten::reshape", "aten::transpose", "aten::t", "aten::select", "aten::slice", "aten::narrow", "aten::detach",
        "aten::alias", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::permute", "aten::size", "aten::stride",
        "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::resize_", "aten::set_", "aten::lift_fresh",
        "aten::unbind", "aten::chunk", "aten::split", "aten::result_type", "aten::view_as", "aten::contiguous",
        "aten::flatten", "aten::unflatten", "aten::numel", "aten::to", "aten::_to_copy"}
count = collections.Counter()
for e in leaves:
    if e.name in SKIP:
        continue
    where = "?"
    for fr in (e.stack or []):
        if ("vqa-attention-networks_amd/" in fr or "bench.py" in fr) and "native_ops.py" not in fr:
            where = fr[fr.find("vqa-attention-networks_amd/") + len("vqa-attention-networks_amd/"):] if "vqa-attention-networks_amd/" in fr else fr[fr.find("bench.py"):]
            break
    shapes = str(e.input_shapes)[:60] if e.input_shapes else ""
    count[(e.name, where, shapes)] += 1
print("%s %s B=%d: aten leaf ops of one train step (views / allocations skipped)" % (args.model, args.dtype, args.batch))
for (name, where, shapes), n in sorted(count.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%3d  %-28s %-70s %s" % (n, name, where, shapes))
