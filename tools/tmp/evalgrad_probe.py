import sys, copy
sys.path[:0] = ["/root/repo/neural-astar_amd", "/root/repo"]
import torch, torch.nn as nn
from neural_astar.planner import NeuralAstar
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30))
for (depth, H, W, mode, rnd) in [(4, 32, 32, "eval", False), (4, 32, 32, "eval", True), (4, 24, 40, "eval", False), (2, 32, 32, "eval", False), (3, 32, 32, "eval", False), (4, 32, 32, "train", False), (4, 64, 64, "eval", False)]:
    torch.manual_seed(depth * 7 + H)
    B = 5
    g = torch.Generator().manual_seed(H + W)
    ref = NeuralAstar(encoder_input="m+", encoder_arch="CNN", encoder_depth=depth, const=3.0)
    with torch.no_grad():
        for m_ in ref.encoder.modules():
            if isinstance(m_, nn.BatchNorm2d):
                m_.weight.uniform_(0.5, 1.5); m_.bias.normal_(0, 0.2)
                m_.running_mean.normal_(0, 0.3); m_.running_var.uniform_(0.5, 2.0)
    na = copy.deepcopy(ref).to(dev)
    ref = ref.double()
    if mode == "eval": na.eval(); ref.eval()
    else: na.train(); ref.train()
    img = torch.rand((B, 1, H, W), generator=g) if rnd else (torch.rand((B, 1, H, W), generator=g) > 0.25).float()
    s = torch.zeros((B, 1, H, W)); s[:, 0, 1, 1] = 1
    gl = torch.zeros((B, 1, H, W)); gl[:, 0, H - 2, W - 2] = 1
    R = torch.randn((B, 1, H, W), generator=g) / (B * H * W)
    # min |relu input| of the reference
    mins = []
    hooks = [m_.register_forward_hook(lambda mod, inp, out: mins.append(float(inp[0].abs().min()))) for m_ in ref.encoder.model if isinstance(m_, nn.ReLU)]
    cost_ref = ref.encode(img.double(), s.double(), gl.double())
    (cost_ref * R.double()).sum().backward()
    cost = na.encode(img.to(dev), s.to(dev), gl.to(dev))
    (cost * R.to(dev)).sum().backward()
    worst = {n: rel(p.grad, q.grad) for (n, p), (_, q) in zip(na.encoder.named_parameters(), ref.encoder.named_parameters()) if float(q.grad.abs().max()) > 0}
    print(depth, H, W, mode, "rnd" if rnd else "bin", na.last_encoder_route, "cost", float((cost.detach().cpu().double() - cost_ref.detach()).abs().max()), "min|relu in|", ["%.1e" % x for x in mins],
          "worst", max(worst.values()), {k: "%.1e" % v for k, v in worst.items() if v > 1e-4}, flush=True)
