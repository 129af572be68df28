import sys; sys.path.insert(0, "/root/repo")
import torch, vqa_amd
fn = vqa_amd.functions.LstmBatchFn
T, B, I, H = 14, 512, 300, 1024
torch.manual_seed(0)
lstm = torch.nn.LSTM(I, H, 1).cuda()
x = torch.randn(T, B, I, device="cuda", requires_grad=True)
def run_hip():
    hs = fn.apply(x, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
    hs.sum().backward()
def run_mi():
    hs, _ = lstm(x); hs.sum().backward()
for name, f in (("hip", run_hip), ("miopen", run_mi)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    print(name, a.elapsed_time(b) / 10, "ms fwd+bwd", flush=True)
vqa_amd.ops.prof_reset(); vqa_amd.ops.prof_enable(True)
run_hip(); torch.cuda.synchronize()
print(vqa_amd.ops.prof_report())
