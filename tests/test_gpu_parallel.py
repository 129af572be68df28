"""Data-parallel path on the GPU: 2 ranks (gloo, both on cuda:0) run MFB on their shard of a global batch
through the stream-overlapped HIP path + GradientAllReducer; the averaged gradients must equal the gradients
of ONE process on the whole batch (MFB samples are independent, mean-CE of equal shards = mean of means;
SURVEY.md 8e).  Exercises the multi-stream bucket synchronisation without needing several GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg():
    import types
    return types.SimpleNamespace(q_vocab_size=50, a_vocab_size=30, emb_dim=24, hidden_dim=64, num_layers=1,
                                 model_name="mfb", glove=False, img_feature_channel=96, img_feature_dim=196)


def _model_and_data(live):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import vqa_amd
    import recipe
    vqa_amd.lib.load()
    cfg = _cfg()
    model = vqa_amd.MFB(cfg)
    model.load_state_dict({k: torch.from_numpy(recipe.weight_for(k, tuple(v.shape), 111))
                           for k, v in model.state_dict().items()})
    model = model.cuda().train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.unit_softmax = not live          # live attention: every tensor gets a non-zero gradient
    N = 8
    img = torch.from_numpy(recipe.img_features(N, 196, 96, 111)).cuda()
    q = torch.from_numpy(recipe.question_tokens(N, 7, 50, 111)).cuda()
    a = torch.from_numpy(recipe.hard_answers(N, 30, 111)).cuda()
    return vqa_amd, model, img, q, a


def _worker(rank, world, port, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    vqa, model, img, q, a = _model_and_data(live=True)
    from importlib import import_module
    par = import_module("vqa-attention-networks_amd.host.parallel")
    par.init_distributed(backend="gloo")
    red = par.GradientAllReducer(model, bucket_bytes=1 << 20)        # several buckets
    lo, hi = par.shard_rows(img.shape[0], rank, world)
    for step in range(2):
        model.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model.forward(img[lo:hi], q[lo:hi]), a[lo:hi])
        loss.backward()
        red.finish()
    torch.cuda.synchronize()
    q_out.put((rank, {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()}, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_full_batch_gradients():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    qo = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, qo)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, grads, nb = qo.get(timeout=300)
        res[rank] = grads
        assert nb > 1
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    vqa, model, img, q, a = _model_and_data(live=True)
    torch.nn.functional.cross_entropy(model.forward(img, q), a).backward()
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        ref = p.grad.detach().cpu()
        for rank in range(world):
            g = torch.from_numpy(res[rank][k])
            assert torch.equal(torch.from_numpy(res[0][k]), g), (k, "replicas disagree after the all-reduce")
            # shard-wise evaluation re-associates the batch reductions.  Well-conditioned tensors: 1e-3 of
            # the norm.  Tensors whose gradient passes through the signed square root of the 196 000 pooled
            # sums (img_conv1d, ques_proj1 and what feeds them) are dominated by the sums nearest 0 and move
            # by O(10 %) under ANY re-association: coarse bound only.
            ill = k.startswith(("img_conv1d", "ques_proj1", "ques_att", "co_att", "lstm", "word_embedding"))
            tol = 0.5 if ill else 1e-3
            assert float((g - ref).norm()) <= tol * float(ref.norm()) + 1e-7, (k, float((g - ref).norm()), float(ref.norm()))
