"""Data-parallel training of the fusion path: one process per GPU, RCCL over xGMI.

Replaces the reference's single-process nn.DataParallel (solver.py:34-36: per
step a parameter broadcast, input scatter, output gather and a reduce-add of
the gradients to GPU 0) by the MI355X-native scheme: replicas never exchange
parameters after the initial broadcast; each step ends with ONE bucketed
all-reduce (average) of the fp32 gradients over RCCL (backend "nccl" on ROCm),
issued on RCCL's own stream as soon as a bucket's gradients exist so that it
overlaps the rest of the backward (the large img_conv1d wgrad comes late).

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is
per-link bound; buckets are therefore few and large (default 64 MiB: MFB's
240 MB of gradients travel in 4 collectives).
"""
import datetime
import os

import torch
import torch.distributed as dist

# Bound on the rendezvous and on every collective of the group (the watchdog of the nccl backend aborts the process when a
# collective exceeds it; gloo raises): a rank that never arrives must end the job with a message, not hold it until an outer
# time limit kills it silently.  The steady-state collectives of this path take milliseconds.
PG_TIMEOUT_S = 120.0


def init_distributed(backend=None, init_method=None, force=False, timeout_s=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank).
    force: create the process group at WORLD_SIZE 1 too (a one-rank RCCL group: the rehearsal a one-GPU box allows).
    timeout_s: bound on the rendezvous and on each collective (default PG_TIMEOUT_S, or the environment's VQF_PG_TIMEOUT_S).

    `init_method` (or the environment variable VQF_DIST_INIT) overrides the env:// rendezvous, e.g.
    "file:///tmp/x/store" — used by the tests, whose parent cannot hold a TCP port open for its children."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    init_method = init_method or os.environ.get("VQF_DIST_INIT") or None
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if timeout_s is None:
            timeout_s = float(os.environ.get("VQF_PG_TIMEOUT_S") or PG_TIMEOUT_S)
        kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(timeout_s)))
        if init_method:
            kw["init_method"] = init_method
        if backend == "nccl":
            local = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
            # RCCL's kernels on a HIGH-PRIORITY stream.  HIP multiplexes its streams onto a few hardware queues per priority
            # (4 by default); measured with rocprofv3 on the one-rank rehearsal (profiles/r03_dp_queues.md): the collective's
            # default-priority stream had landed on the SAME hardware queue as the compute stream, so every all-reduce kernel
            # ran serialised with the backward instead of beside it.  A different priority is a different hardware queue.
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kw["pg_options"] = opts
            except Exception:                     # an older torch without the option object: default stream priority
                pass
        try:
            dist.init_process_group(backend, **kw)
        except Exception as e:
            raise RuntimeError("rank %d of %d: torch.distributed rendezvous (%s, %s) did not complete within %.0f s -- a peer rank "
                               "never arrived or the store is unreachable: %s: %s"
                               % (rank, world, backend, init_method or "env://%s:%s" % (os.environ.get("MASTER_ADDR"),
                                                                                         os.environ.get("MASTER_PORT")),
                                  float(timeout_s), type(e).__name__, str(e)[:300])) from e
    return rank, world, local


# Launch form of the large-tile GEMMs while ANY reducer is alive: set by the first one, restored by the last close().  Kept
# here, not per reducer: a second reducer built while the first is alive (or `red = GradientAllReducer(m)` rebinding, where the
# new object exists before the old one's __del__ runs) must neither mistake the first one's value for the user's choice nor have
# the old one's close() restore the persistent form underneath it.
_GEMM_FORM = {"count": 0, "saved": {}}
_GEMM_FORM_OPTIONS = ("gemm_f32_persist", "gemm_bf16_persist")


def _gemm_form_acquire(ops, gemm_workgroups):
    want = 1 if gemm_workgroups == "persistent" else 0
    saved = _GEMM_FORM["saved"]
    for name in _GEMM_FORM_OPTIONS:
        if name in saved:                       # written by a live reducer, not by the user
            if gemm_workgroups is not None:
                ops.set_option(name, want)
            continue
        if gemm_workgroups is None and ops.get_option(name) >= 0:
            continue                            # set explicitly by the user: respected
        saved[name] = ops.set_option(name, want)
    _GEMM_FORM["count"] += 1


def _gemm_form_release(ops):
    _GEMM_FORM["count"] -= 1
    if _GEMM_FORM["count"] <= 0:
        _GEMM_FORM["count"] = 0
        for name, prev in _GEMM_FORM["saved"].items():
            ops.set_option(name, prev)
        _GEMM_FORM["saved"].clear()


def shard_rows(n_global, rank, world):
    """Rows [lo, hi) of the global minibatch owned by `rank` (contiguous row blocks)."""
    per = n_global // world
    rem = n_global % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


class GradientAllReducer:
    """Bucketed, overlapped gradient averaging for a replica of `module`.

    usage per step:
        opt.zero_grad(set_to_none=True); loss.backward(); reducer.finish(); opt.step()
    After finish(), every p.grad is a view into a flat bucket holding the average
    over ranks.  With world_size == 1 it is a no-op (grads are left untouched).
    A parameter that received no gradient on ANY rank in the first step (HieCoAtten's fc_Wbq: hieCoAtten.py:11 builds it,
    :31 never uses it) keeps p.grad = None, as without the reducer -- so it gets no optimizer state either -- and leaves
    the buckets; the graph is taken to be static after that (such a parameter receiving a gradient later raises).  A parameter
    with a gradient on SOME ranks only gets the average over all ranks (absent = zero) through buckets every rank launches
    from finish(); the first step runs its collectives in strict bucket order, so ranks never pair different buckets.
    gemm_workgroups: None (default: one workgroup per tile at world > 1 unless the library option was set explicitly),
    "per-tile" or "persistent" -- how the large-tile GEMMs launch while this reducer is alive (see __init__).
    """

    def __init__(self, module, bucket_bytes=64 << 20, process_group=None, broadcast=True, gemm_workgroups=None,
                 single_rank=False):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # single_rank: run the buckets, hooks and collectives in a ONE-rank group as well (the average over one rank is the
        # identity): exercises the RCCL stream semantics on a one-GPU box
        self.active = self.world > 1 or (single_rank and dist.is_initialized())
        self.params = [p for p in module.parameters() if p.requires_grad]
        if self.active and broadcast:
            self.broadcast_parameters()
        self.buckets = []       # list of dict(flat, params=[(p, offset, numel)], pending, handle)
        self._index = {}
        # timing = True: finish() brackets its waits with events on the compute stream; exposed_ms() then
        # reports, per bucket, how long after the last backward kernel that bucket's collective completed
        # (the part of the all-reduce the backward did NOT hide)
        self.timing = False
        self._marks = []
        self._holds_gemm_form = False
        self._hooks = []
        self._seen = set()          # id() of the parameters whose hook fired in the current step
        self.unused = None          # id() of the parameters without a gradient on any rank (known after the first finish())
        self.ragged = None          # ... with a gradient on some ranks only (their buckets are launched from finish())
        self._next = 0
        self._bucket_bytes = bucket_bytes
        if self.active:
            # The large-tile GEMMs normally run as PERSISTENT workgroups (one per CU, holding all of its LDS for the whole
            # launch: csrc/gemm_f32_big.hip).  The collective's kernels could then not get onto a CU before the 14-ms
            # weight-gradient GEMM they are meant to overlap with has ended.  Data-parallel runs therefore launch one
            # workgroup per tile (a CU frees up every ~0.5 ms; costs the GEMMs ~0.7 %) unless the option was set explicitly
            # (vqf_set_option / the environment at load time).  An explicit library option, not an environment mutation:
            # close() (or leaving the `with` block, or the reducer's collection) restores what was there.
            if gemm_workgroups is not None and gemm_workgroups not in ("per-tile", "persistent"):
                raise ValueError("gemm_workgroups: 'per-tile', 'persistent' or None")
            from . import ops, lib as _l
            try:
                _gemm_form_acquire(ops, gemm_workgroups)
                self._holds_gemm_form = True
            except _l.VqfError:
                # library not built: only a CPU rehearsal of the reducer (gloo) may go on without it
                if any(p.is_cuda for p in self.params):
                    raise
            self._build_buckets(bucket_bytes)
            self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def close(self):
        """Remove the gradient hooks and restore the library options this reducer changed."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._holds_gemm_form:
            from . import ops
            self._holds_gemm_form = False
            _gemm_form_release(ops)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gemm_workgroups(self):
        """{'f32': ..., 'bf16': ...}: how the large-tile GEMMs of each family launch right now."""
        out = {}
        try:
            from . import ops
            for fam, name in (("f32", "gemm_f32_persist"), ("bf16", "gemm_bf16_persist")):
                out[fam] = "one per tile" if ops.get_option(name) == 0 else "persistent, one per CU"
        except Exception as e:      # no library (CPU-only rehearsal of the reducer)
            out = {"f32": "n/a", "bf16": "n/a", "note": str(e)[:80]}
        return out

    # -- setup ---------------------------------------------------------------
    def broadcast_parameters(self, src=0):
        """identical replicas: rank `src`'s parameters and buffers go to everyone (once)."""
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src=src, group=self.group)

    def _build_buckets(self, bucket_bytes):
        """Buckets in the order the gradients become ready (roughly reverse registration order).  Parameters without a gradient
        on any rank (`unused`) hold no bucket space; parameters that received a gradient on SOME ranks only (`ragged`, known
        after the first step) go into buckets of their own at the end, flagged `late`: those are launched from finish() only."""
        self.buckets, self._index = [], {}
        live = [p for p in reversed(self.params) if not (self.unused and id(p) in self.unused)]
        eager = [p for p in live if not (self.ragged and id(p) in self.ragged)]
        late = [p for p in live if self.ragged and id(p) in self.ragged]
        for ps_all, is_late in ((eager, False), (late, True)):
            cur, cur_bytes, groups = [], 0, []
            for p in ps_all:
                nbytes = p.numel() * p.element_size()
                if cur and cur_bytes + nbytes > bucket_bytes:
                    groups.append(cur)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nbytes
            if cur:
                groups.append(cur)
            for ps in groups:
                bi = len(self.buckets)
                total = sum(p.numel() for p in ps)
                flat = torch.zeros(total, dtype=ps[0].dtype, device=ps[0].device)
                entries, off = [], 0
                for p in ps:
                    entries.append((p, off, p.numel()))
                    self._index[id(p)] = (bi, off)
                    off += p.numel()
                self.buckets.append(dict(flat=flat, params=entries, pending=len(entries), handle=None, late=is_late,
                                         complete=False))
        self._next = 0              # first step only: the next bucket (by index) the strict order may launch

    # -- per step ------------------------------------------------------------
    # Collective ORDER must be the same on every rank.  Steady state: a bucket is launched the moment its last gradient has
    # arrived; with the same autograd graph on every rank that order is the same everywhere.  What may differ between ranks is
    # WHICH parameters receive a gradient (a branch one shard does not take), so
    #   * the FIRST step launches strictly in bucket-index order (complete buckets as early as the order allows, the rest from
    #     finish()): the same sequence on every rank whatever arrived where;
    #   * finish() of that step then all-reduces (SUM) a has-gradient mask: parameters seen by no rank leave the buckets and
    #     keep grad = None; parameters seen by some ranks only move to `late` buckets, which every rank launches from
    #     finish(), in index order, after the eager ones (zero-filled where the gradient is absent);
    #   * a parameter that had a gradient on every rank in step one and misses it later would make this rank launch its bucket
    #     out of order: that raises here instead of mis-pairing collectives (the peers then end on the group's timeout).
    def _on_grad(self, p):
        if id(p) not in self._index:
            raise RuntimeError("GradientAllReducer: a parameter that had no gradient on any rank in the first step (shape %s) "
                               "received one now; the reducer assumes a static graph -- build a new reducer" % (tuple(p.shape),))
        self._seen.add(id(p))
        bi, off = self._index[id(p)]
        b = self.buckets[bi]
        # The gradient is only NOTED here; the bucket is filled by ONE multi-tensor copy when its last gradient has arrived
        # (a copy kernel per parameter put ~100 small launches on the backward's critical path: 0.3 ms of the headline step).
        b.setdefault("ready", []).append((b["flat"][off:off + p.numel()], p.grad.reshape(-1)))
        if p.is_cuda:
            # gradients of one bucket may be produced on different streams (the image projection and
            # its weight gradient run on a side stream): remember where each one was produced
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(p.device))
            b.setdefault("events", []).append(ev)
        b["pending"] -= 1
        if b["pending"] == 0:
            b["complete"] = True
            if self.unused is None:               # first step: strict index order
                self._launch_in_order()
            elif not b["late"]:
                self._launch(b)

    def _launch_in_order(self):
        while self._next < len(self.buckets) and self.buckets[self._next]["complete"]:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b):
        cur = torch.cuda.current_stream(b["flat"].device) if b["flat"].is_cuda else None
        if b.get("events"):
            for ev in b["events"]:
                cur.wait_event(ev)             # the fill (and the collective behind it) is ordered after EVERY gradient
            b["events"] = []
        ready = b.get("ready") or []
        if ready:
            if b["flat"].is_cuda:
                # A gradient may have been allocated on another stream (the side stream's dwi, functions.ImgProjLateFn) and is
                # freed in finish(), when p.grad is re-pointed at the bucket: tell the caching allocator that THIS stream reads
                # it, so that its block is not handed out again under the in-flight copy
                for _, g in ready:
                    g.record_stream(cur)
            torch._foreach_copy_([d for d, _ in ready], [g for _, g in ready])
            b["ready"] = []
        b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.AVG if self._has_avg() else dist.ReduceOp.SUM,
                                      group=self.group, async_op=True)

    def _has_avg(self):
        return dist.get_backend(self.group) == "nccl"

    def _launch_rest(self, first):
        """finish(): every bucket that is not in flight yet, in index order; gradients absent on THIS rank contribute zeros."""
        for b in self.buckets:
            if b["handle"] is not None:
                continue
            if not first and not b["late"] and b["pending"] != 0:
                missing = [tuple(p.shape) for p, _, _ in b["params"] if id(p) not in self._seen]
                raise RuntimeError("GradientAllReducer: parameter(s) of shape %s received a gradient on every rank in the first "
                                   "step and none on this rank now: launching their bucket here would pair it with another "
                                   "bucket's collective on the peers.  The reducer assumes a static graph after the first step "
                                   "-- build a new reducer" % (missing,))
            for p, off, n in b["params"]:
                if id(p) not in self._seen:
                    b["flat"][off:off + n].zero_()
            self._launch(b)
        self._next = len(self.buckets)

    def finish(self):
        """wait for the collectives and expose the averaged gradients as p.grad."""
        if not self.active:
            return
        first = self.unused is None
        self._launch_rest(first)                   # BEFORE the mask exchange: the same collective sequence on every rank
        if first:
            self.unused, self.ragged = self._participation()
        cuda = self.timing and self.buckets and self.buckets[0]["flat"].is_cuda
        marks = None
        if cuda:
            cur = torch.cuda.current_stream(self.buckets[0]["flat"].device)
            marks = [torch.cuda.Event(enable_timing=True)]
            marks[0].record(cur)               # behind the last backward kernel of the compute stream
        for b in self.buckets:
            b["handle"].wait()
            if cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(cur)                 # the compute stream now waits on / has seen this collective
                marks.append(ev)
            if not self._has_avg():
                b["flat"].div_(self.world)
            for p, off, n in b["params"]:
                if id(p) not in self.unused:
                    p.grad = b["flat"][off:off + n].view_as(p)
            b["pending"] = len(b["params"])
            b["handle"], b["complete"] = None, False
        self._seen = set()
        self._next = 0
        if first and (self.unused or self.ragged):
            # the buckets of the following steps: live gradients only, ragged ones at the end (the views handed out above keep
            # this step's buffers alive)
            self._build_buckets(self._bucket_bytes)
        if marks:
            self._marks.append(marks)
            if len(self._marks) > 1024:                # timing is a bench-only switch; never grow without bound
                del self._marks[:512]

    def _participation(self):
        """-> (unused, ragged): ids of the parameters whose hook fired on NO rank / on SOME BUT NOT ALL ranks in the step that just
        ran.  One small SUM all-reduce of a has-gradient mask, once (its result is read on the host: a synchronisation the
        steady state does not pay); issued after every bucket of the step, so every rank reaches it at the same place."""
        dev = self.params[0].device if self.params else torch.device("cpu")
        mask = torch.tensor([1 if id(p) in self._seen else 0 for p in self.params], dtype=torch.int32, device=dev)
        if mask.numel():
            dist.all_reduce(mask, op=dist.ReduceOp.SUM, group=self.group)
        counts = mask.tolist()
        unused = {id(p) for p, m in zip(self.params, counts) if m == 0}
        ragged = {id(p) for p, m in zip(self.params, counts) if 0 < m < self.world}
        return unused, ragged

    def exposed_ms(self):
        """Mean over the timed steps of [ms from the end of the backward kernels to bucket i's completion],
        in launch order (bucket 0 = the gradients produced first); the last entry is the whole exposed tail."""
        if not self._marks:
            return []
        torch.cuda.synchronize()
        marks_all, self._marks = self._marks, []      # a bench-only switch: the window is consumed here, nothing accumulates
        n = min(len(m) for m in marks_all) - 1
        acc = [0.0] * n
        for marks in marks_all:
            for i in range(n):
                acc[i] += marks[0].elapsed_time(marks[i + 1])
        return [round(a / len(marks_all), 4) for a in acc]

    def bucket_bytes_list(self):
        return [b["flat"].numel() * b["flat"].element_size() for b in self.buckets]

    def gradient_bytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)
