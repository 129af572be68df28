"""A stalled rank must end a bare `python bench.py --gpus N` launch with a diagnosis, not with silence (VERDICT r04 next #1;
the reference's only multi-GPU mechanism is nn.DataParallel, solver.py:34-36 -- its replacement is one process per GPU, and a
rank that never reaches the rendezvous on the first 8-GPU lease would otherwise hold the job until an outer time limit kills it).

The stall happens BEFORE torch.distributed is initialised, i.e. before anything touches the GPU: these tests run on the CPU
(and again on the GPU box).  One attempt each, no retry."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_STRIP = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VQF_DIST_INIT", "VQF_PG_TIMEOUT_S")


def _run(extra, stall_rank=1, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in _STRIP}
    env.update(VQF_TEST_STALL_RANK=str(stall_rank), VQF_TEST_STALL_S="400")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "gloo", "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=timeout,
                       cwd=ROOT, env=env)
    return p, time.time() - t0


def test_a_rank_that_never_reaches_the_rendezvous_trips_the_launch_deadline():
    """Rank 1 sleeps before init_distributed; the parent's deadline (20 s here, 420 s by default) expires: exit code 124, the
    message names rank 1 and quotes its last progress marker, both rank logs are printed and kept under gpurun_out/."""
    p, dt = _run(["--launch-timeout", "20"])
    err = p.stderr
    assert p.returncode == 124, (p.returncode, err[-3000:])
    assert dt < 90, dt                                                        # inside the deadline + kill grace, not the 400 s stall
    assert "launch deadline: rank(s) [0, 1] had not reached 'process group ready'" in err, err[-3000:]
    assert "rank(s) [1] never ARRIVED at the rendezvous, rank(s) [0] were waiting in it for their peers" in err, err[-3000:]
    assert "rank 1: NEVER reached 'process group ready'" in err and "VQF_TEST_STALL_RANK: sleeping" in err
    assert "rank 0: NEVER reached" in err and "rendezvous: init_process_group(gloo)" in err      # rank 0 waits in the rendezvous
    assert "last 40 lines of gpurun_out/rank0.log" in err and "last 40 lines of gpurun_out/rank1.log" in err
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]                    # no half result
    for r in (0, 1):
        assert os.path.exists(os.path.join(ROOT, "gpurun_out", "rank%d.log" % r))


def test_the_rendezvous_bound_names_the_missing_rank():
    """The same stall with the process-group bound (host/parallel.py: 120 s by default, 8 s here) shorter than the parent's
    deadline: rank 0's init_process_group gives up, says which peers never got to the rendezvous (their last markers), exits
    non-zero; the parent ends rank 1 and relays both logs."""
    p, dt = _run(["--launch-timeout", "150", "--pg-timeout", "8"])
    err = p.stderr
    assert p.returncode not in (0, 124), (p.returncode, err[-3000:])
    assert dt < 120, dt
    assert "rank(s) [0] ended non-zero" in err and "while rank(s) [1] were still running" in err, err[-3000:]
    assert "did not complete within 8 s" in err                                # rank 0's own message, from its log
    assert "rank 1 last marker" in err and "VQF_TEST_STALL_RANK: sleeping" in err
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]


def test_in_rank_watchdog_ends_a_rank_blocked_in_the_rendezvous(tmp_path):
    """Launches this file does not supervise (the driver's torchrun line): every rank arms a GIL-free timer for the launch
    phase.  Rank 0 of a 2-rank world whose peer is never started blocks in the rendezvous (bound 300 s here); the watchdog
    (6 s) dumps the stacks and ends the process, the stage file says where it was."""
    env = {k: v for k, v in os.environ.items() if k not in _STRIP}
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", VQF_DIST_INIT="file://" + str(tmp_path / "store"))
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "gloo", "--no-cpu-baseline", "--launch-timeout", "6", "--pg-timeout", "300"],
                       capture_output=True, text=True, timeout=200, cwd=ROOT, env=env)
    assert p.returncode != 0 and time.time() - t0 < 90, (p.returncode, p.stderr[-2000:])
    assert "Timeout (0:00:06)!" in p.stderr and "init_process_group" in p.stderr, p.stderr[-2000:]
    assert "rendezvous: init_process_group(gloo)" in open(os.path.join(ROOT, "gpurun_out", "rank0.stage")).read()
