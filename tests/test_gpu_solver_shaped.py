"""The drop-in modules inside a loop shaped like the reference's Solver (solver.py:52-117, 119-182,
184-190): Adam(lr=7e-4), CE / KLDiv criterion, `self.model.forward(i, q)` called positionally,
loss.backward(), optimizer.step(), argmax accuracy, state_dict save -> load round trip."""
import io
import types

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cfg(name):
    return types.SimpleNamespace(q_vocab_size=60, a_vocab_size=16, emb_dim=24, hidden_dim=64, num_layers=1,
                                 model_name=name, glove=False, img_feature_channel=96, img_feature_dim=196,
                                 lr=7e-4)


@pytest.mark.parametrize("tail", ["torch", "hip"])
@pytest.mark.parametrize("name", ["mfb", "mhb_coAtt", "mhb"])
def test_training_loop_learns_and_checkpoints(name, tail):
    """tail = "torch": the solver's own criterion / optimizer objects (nn.CrossEntropyLoss / nn.KLDivLoss, torch.optim.Adam,
    solver.py:25-29) around the drop-in modules; "hip": the path's own (vqa_amd.train_step: same names and signatures).
    Adam at cfg.lr = 7e-4 (cfg.py:18) with the solver's decay step (solver.py:47-50)."""
    import vqa_amd
    vqa_amd.lib.load()
    cfg = _cfg(name)
    torch.manual_seed(0)
    cls = {"mfb": vqa_amd.MFB, "mhb_coAtt": vqa_amd.MHBCoAtt, "mhb": vqa_amd.MHB}[name]
    model = cls(cfg)
    for n, p in model.named_parameters():                       # train_models.py:54-56
        if n.find('bias') == -1:
            torch.nn.init.xavier_uniform_(p)
    model.to("cuda:0")
    if tail == "torch":
        criterion = torch.nn.KLDivLoss() if name in ("mhb_coAtt", "mhb") else torch.nn.CrossEntropyLoss()   # solver.py:26-29
        optimizer = torch.optim.Adam(model.parameters(), lr=cfg.lr)                                          # solver.py:30
    else:
        criterion = vqa_amd.train_step.criterion_for(name)
        optimizer = vqa_amd.Adam(model.parameters(), lr=cfg.lr)
    N, T = 8, 9
    g = torch.Generator().manual_seed(1)
    i = torch.relu(torch.randn((N, 196, 96), generator=g)).cuda()
    q = torch.randint(1, 60, (N, T), generator=g).cuda()
    q_l = torch.full((N,), T, dtype=torch.long).cuda()
    hard = torch.randint(0, 16, (N,), generator=g).cuda()
    a = F.one_hot(hard, 16).float() if name != "mfb" else hard
    model.train()
    losses = []
    lr = cfg.lr
    for step in range(160):                                     # solver.py:68-94
        if step == 120:                                         # solver.py:47-50 (update_lr)
            lr *= 0.5
            for param_group in optimizer.param_groups:
                param_group['lr'] = lr
        logits = model.forward(i, q, q_l) if name == "mhb" else model.forward(i, q)
        loss = criterion(logits, a)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        losses.append(float(loss))
    assert losses[-1] < 0.9 * losses[0], (losses[0], losses[-1])     # memorises 8 samples
    pred = F.softmax(logits, dim=1).max(1)[1]                   # solver.py:96-101
    acc = (pred == hard).float().mean()
    assert 0.0 <= float(acc) <= 1.0
    # solver.save(): torch.save(clean_state_dict(model.state_dict())) -> train_models.py:58-60 load
    buf = io.BytesIO()
    torch.save(model.state_dict(), buf)
    buf.seek(0)
    clone = cls(cfg)
    clone.load_state_dict(torch.load(buf))
    clone.to("cuda:0").eval()
    model.eval()                                                # solver.val(): self.model.eval()
    with torch.no_grad():
        a1 = model.forward(i, q, q_l) if name == "mhb" else model.forward(i, q)
        a2 = clone.forward(i, q, q_l) if name == "mhb" else clone.forward(i, q)
    assert torch.equal(a1, a2)
