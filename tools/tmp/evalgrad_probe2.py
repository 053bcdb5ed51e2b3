import sys, copy
sys.path[:0] = ["/root/repo/neural-astar_amd", "/root/repo"]
import torch, torch.nn as nn
from neural_astar.planner import NeuralAstar
import neural_astar.encoder_train as ET
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30))
for (depth, H, W, B, co1, prec) in [(4, 32, 32, 5, True, "f16x3"), (4, 32, 32, 5, False, "f16x3"), (4, 32, 32, 8, True, "f16x3"), (4, 32, 32, 5, True, "f16"), (4, 32, 64, 5, True, "f16x3"), (4, 16, 64, 5, True, "f16x3")]:
    ET.CO1_STREAMS = co1
    torch.manual_seed(depth * 7 + H)
    g = torch.Generator().manual_seed(H + W)
    ref = NeuralAstar(encoder_input="m+", encoder_arch="CNN", encoder_depth=depth, const=3.0)
    with torch.no_grad():
        for m_ in ref.encoder.modules():
            if isinstance(m_, nn.BatchNorm2d):
                m_.weight.uniform_(0.5, 1.5); m_.bias.normal_(0, 0.2)
                m_.running_mean.normal_(0, 0.3); m_.running_var.uniform_(0.5, 2.0)
    na = copy.deepcopy(ref).to(dev).eval()
    na.encoder_backend = "hip_" + prec
    ref = ref.double().eval()
    img = (torch.rand((B, 1, H, W), generator=g) > 0.25).float()
    s = torch.zeros((B, 1, H, W)); s[:, 0, 1, 1] = 1
    gl = torch.zeros((B, 1, H, W)); gl[:, 0, H - 2, W - 2] = 1
    R = torch.randn((B, 1, H, W), generator=g) / (B * H * W)
    cost_ref = ref.encode(img.double(), s.double(), gl.double())
    (cost_ref * R.double()).sum().backward()
    cost = na.encode(img.to(dev), s.to(dev), gl.to(dev))
    (cost * R.to(dev)).sum().backward()
    worst = {n: rel(p.grad, q.grad) for (n, p), (_, q) in zip(na.encoder.named_parameters(), ref.encoder.named_parameters()) if float(q.grad.abs().max()) > 0}
    print(depth, H, W, "B", B, "co1", co1, prec, na.last_encoder_route, "worst", "%.1e" % max(worst.values()), {k: "%.1e" % v for k, v in worst.items() if v > 1e-4 and ("10" in k or "9" in k)}, flush=True)
