"""bench.py keeps its output contract: ONE JSON line with the driver's keys, the roofline object of the dominant
kernel and the cpu_baseline object (a short run: 2 timed steps; the CPU leg is the bounded sample itself)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--cpu-batch", "64",           # the default (the metric's B=512) takes ~75 s of CPU time
                        "--secondary-steps", "2", "--secondary-warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                 ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] in ("TFLOP/s", "GB/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    assert "traffic" in r                                   # bytes from the committed PMC passes, or null
    assert r["traffic"] is None or str(r["traffic_source"]).startswith("profiles/")
    assert r["wgrad"]["frac"] > 0.3 and "ZERO" in r["wgrad"]["operands"]      # faithful MFB: dP == 0, said so
    assert 0.3 < r["wgrad_live"]["frac"] < 1.0                                   # the same launch on live operands
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["config1"]["value"] > 0 and "B=32" in c["config1"]["sample"]
    assert abs(d["value"] - 512 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3
    cfg = d["config"]
    assert cfg["ranks_seen"] == 1 and cfg["allreduce_exposed_ms"] == [] and "one compute stream" in cfg["streams"]
    assert cfg["gemm_workgroups"].startswith("persistent")
    assert cfg["gemm_workgroups_by_family"] == {"f32": "persistent, one per CU", "bf16": "persistent, one per CU"}
    assert "with_mask_generation" in c and 0 < c["with_mask_generation"]["value"] <= c["value"] * 1.05
    assert 0.3 < d["step_roofline"]["frac"] < 1.0
    # HBM-bound kernels of the step: algorithmic GB/s now, counter bytes when a committed PMC pass exists
    for k in ("mfb_fuse_fwd", "mfb_fuse_bwd", "glimpse_pool_fwd", "glimpse_pool_bwd", "att_logits_bwd"):
        h = d["roofline_hbm_kernels"][k]
        assert 0.05 < h["frac"] < 1.0 and "traffic" in h and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-3
    assert "scale_rows" not in d["roofline_hbm_kernels"] and "rowdot" not in d["roofline_hbm_kernels"]   # folded into co_att_conv1
    # what plain streaming kernels of THIS library reach on the chip, same units (the guide: 6.29 TB/s copy, 6.0 read sweep)
    for k in ("copy", "copy_nt", "read_sweep", "read_sweep_nt", "torch_copy"):
        y = d["hbm_yardsticks"][k]
        assert 0.3 < y["frac"] < 1.0 and abs(y["frac"] - y["achieved"] / 8000.0) < 1e-3
    # the line stays small enough for the driver's record, and ends with the summary of BASELINE configs 3 and 4
    assert len(lines[0]) < 6500, len(lines[0])
    assert lines[0].rstrip().endswith("}}}") and lines[0].index('"secondary_summary"') > len(lines[0]) - 2000
    assert "kernels_ms_per_step" not in d and "secondary" not in d and d["launches_per_step"] > 50 and d["top_kernels_ms_per_step"]
    # ... the full objects (census tables, notes, the secondary configs' roofline objects) are in the census file
    full = json.load(open(os.path.join(ROOT, d["census_file"])))
    assert full["value"] == d["value"] and full["kernels_ms_per_step"] and "kernels_note" in full
    sec, summ = full["secondary"], d["secondary_summary"]
    for which, model, batch, dtype, peak in (("config3", "mhb_coAtt", 512, "bf16", 2500.0), ("config4", "hieCoAtten", 256, "f32", 157.3),
                                             ("config3_all", "mhb_coAtt", 512, "bf16", 2500.0)):
        s2, sm = sec[which], summ[which]
        assert "error" not in s2, s2
        assert s2["dtype"] == dtype and s2["steps"] == 2 and model in s2["metric"] and str(batch) in s2["metric"]
        assert abs(s2["value"] - batch * 1000.0 / s2["ms_per_step"]) / s2["value"] < 1e-3
        r2 = s2["roofline"]
        assert r2 is not None and r2["bound"] == "mfma" and r2["peak"] == peak and r2["launches"] == 2
        assert abs(r2["frac"] - r2["achieved"] / r2["peak"]) < 1e-3 and 0.05 < r2["frac"] < 1.0
        assert r2["wgrad"]["frac"] > 0.05 and "workload" in s2["config"] and s2["kernels_ms_per_step"] and "kernels_note" in s2
        if model == "mhb_coAtt":
            assert "two streams" in s2["config"]["streams"] and "128 CUs" in s2["config"]["streams"]
        assert sm["ms_per_step"] == s2["ms_per_step"] and sm["value"] == s2["value"] and sm["dtype"] == dtype
        assert sm["roofline_frac"] == r2["frac"] and sm["roofline_peak"] == peak and sm["step_roofline_frac"] == s2["step_roofline"]["frac"]
        assert sm["launches_per_step"] > 10
    # the data-parallel machinery in a one-rank RCCL group (solver.py:34-36 replaced), timed in the same run: what a rank pays
    # before any byte crosses xGMI
    dp, dps = sec["dp_one_rank"], summ["dp_one_rank"]
    assert "error" not in dp, dp
    assert dps["backend"] == "nccl" and dps["gemm_workgroups"] == "one per tile" and dps["steps"] == 2
    assert dps["ms_per_step"] == dp["ms_per_step"] and 0.5 * d["ms_per_step"] < dps["ms_per_step"] < 2.0 * d["ms_per_step"]
    assert dps["plain_ms_per_step"] == d["ms_per_step"] and abs(dps["overhead_frac"] - (dps["ms_per_step"] / d["ms_per_step"] - 1)) < 1e-3
    assert len(dps["allreduce_exposed_ms"]) == len(dp["allreduce_bucket_bytes"]) >= 4 and sum(dp["allreduce_bucket_bytes"]) == 240124080


def test_stalled_rank_trips_the_launch_deadline_on_the_gpu_box():
    """tests/test_launch_deadline.py's first case again where the driver's GPU suite runs (the stall precedes any GPU call)."""
    import test_launch_deadline as t
    t.test_a_rank_that_never_reaches_the_rendezvous_trips_the_launch_deadline()


def test_bench_refuses_an_inconsistent_explicit_world_size():
    """An explicit WORLD_SIZE that differs from --gpus must not silently run the wrong rank count."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def _check_two_rank_line(d):
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 1024
    assert d["config"]["grad_allreduce_bytes"] == 240124080
    assert d["config"]["ranks_seen"] == 2 and d["config"]["backend"] == "gloo"
    assert d["config"]["gemm_workgroups"].startswith("one per tile")      # host/parallel.py: the collective needs CUs
    assert d["config"]["gemm_workgroups_by_family"] == {"f32": "one per tile", "bf16": "one per tile"}
    assert "secondary_summary" not in d                                     # configs 3 / 4 ride on the 1-GPU line only
    assert sum(d["config"]["allreduce_bucket_bytes"]) == 240124080
    ex = d["config"]["allreduce_exposed_ms"]
    assert len(ex) == len(d["config"]["allreduce_bucket_bytes"]) and all(b >= a - 1e-3 for a, b in zip(ex, ex[1:]))
    assert abs(d["value"] - 1024 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3
    assert "cpu_baseline" not in d
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.2 < r["frac"] < 1.0


def test_bench_bare_gpus_2_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with NO torchrun environment (how the driver's bench command is spelt): the process
    starts two fresh child ranks itself (torch.distributed.run as a child, before anything has touched the GPU), relays rank
    0's single JSON line and exits 0.  gloo, both ranks on cuda:0 (solver.py:34-36's nn.DataParallel replaced)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "VQF_DIST_INIT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["steps"] == 3
    _check_two_rank_line(d)


def test_bench_two_ranks_on_one_gpu_gloo_rehearsal(tmp_path):
    """The N>1 code path of bench.py end to end without a second GPU: two fresh child processes with the
    torchrun-style environment (started before anything touches the GPU), backend gloo, both on cuda:0.
    Checks the contract line of rank 0: n_gpus, the gradient payload (MFB: 60 031 020 fp32 parameters =
    240 124 080 bytes), ranks seen by torch.distributed, the per-bucket exposed all-reduce times."""
    env = dict(os.environ, WORLD_SIZE="2", LOCAL_RANK="0", VQF_DIST_INIT="file://" + str(tmp_path / "store"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--no-cpu-baseline"]
    procs = [subprocess.Popen(cmd, cwd=ROOT, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r][1][-3000:])
    lines = [l for l in outs[0][0].splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.strip().startswith("{")]
    d = json.loads(lines[0])
    _check_two_rank_line(d)


def test_bench_bf16_cli_run_carries_a_roofline():
    """`bench.py --model mhb_coAtt --dtype bf16` (round 2 printed "roofline": null for every bf16 run): the bf16 image-projection
    launch against the dense bf16 MFMA peak, its weight gradient, and no secondary block on a non-headline run."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "mhb_coAtt", "--dtype", "bf16", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith("{")][0])
    r = d["roofline"]
    assert d["dtype"] == "bf16" and r is not None and r["peak"] == 2500.0 and r["launches"] == 2 and 0.1 < r["frac"] < 1.0
    assert "gemm_bf16" in r["kernel"] and r["wgrad"]["frac"] > 0.1 and "secondary_summary" not in d
    assert "mhb_coAtt" in d["metric"] and d["config"]["workload"].startswith("mhb_coAtt")
