"""Dev probe (GPU box): gradients w.r.t. every block output of the U-Net, HIP training path vs the torch module in float64."""
import copy
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT, os.path.join(ROOT, "tests")]
import test_unet_gpu as TU  # noqa: E402
from neural_astar import encoder_hip as E  # noqa: E402
from neural_astar.planner import NeuralAstar  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
B = 3
pr = syn.random_obstacle_maps(B, 32, 32, 0.25, seed=4)
m, s, g = (torch.from_numpy(x) for x in pr)
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ref = NeuralAstar(encoder_arch="Unet", encoder_depth=DEPTH)
ref.encoder = TU._calibrated_unet(depth=DEPTH, seed=3)
na = copy.deepcopy(ref).to(dev).train()
ref = ref.double().train()
R = torch.randn((B, 1, 32, 32), generator=torch.Generator().manual_seed(9)) / (B * 1024)
# reference: gradient w.r.t. the output of every conv-BN-ReLU block = grad_output of its ReLU
plan = E.unet_layer_plan(ref.encoder.model)
relu_of = {}
model = ref.encoder.model
conv_to_name = {id(st[4]): st[1] for st in plan if st[0] == "conv"}
mods = []
for mod in model.modules():
    if isinstance(mod, nn.Sequential):
        ch = list(mod)
        for i, c in enumerate(ch):
            if isinstance(c, nn.Conv2d) and id(c) in conv_to_name:
                for j in range(i + 1, min(i + 3, len(ch))):
                    if isinstance(ch[j], nn.ReLU):
                        relu_of[conv_to_name[id(c)]] = ch[j]
refg = {}
for name, r in relu_of.items():
    r.inplace = False
    r.register_full_backward_hook(lambda mod, gi, go, name=name: refg.__setitem__(name, go[0].detach().clone()))
cost_ref = ref.encode(m.double(), s.double(), g.double())
(cost_ref * R.double()).sum().backward()
dbg = {}
na.encoder._nastar_debug = dbg
na.encoder_backend = "hip_f16x3"
cost = na.encode(m.to(dev), s.to(dev), g.to(dev))
(cost * R.to(dev)).sum().backward()
torch.cuda.synchronize()
print("cost err", float((cost.detach().cpu().double() - cost_ref.detach()).abs().max()))
for (n1, p1), (_, p2) in zip(na.encoder.named_parameters(), ref.encoder.named_parameters()):
    if p1.grad is not None and float(p2.grad.abs().max()) > 1e-12:
        print(f"   param {n1:45s} {float((p1.grad.cpu().double() - p2.grad).abs().max() / p2.grad.abs().max()):.1e}")
for st in plan:
    if st[0] != "conv" or st[1] not in dbg or st[1] not in refg:
        continue
    buf, S, (b, h, w, C) = dbg[st[1]]
    o = buf.view(torch.float16).float().reshape(b, h, w, 2 * C).cpu()
    got = ((o[..., :C] + o[..., C:]) / float(S)).permute(0, 3, 1, 2).double()
    r = refg[st[1]]
    err = float((got - r).abs().max() / r.abs().max())
    ratio = float((got * r).sum() / (r * r).sum())
    print(f"{st[1]:6s} C={C:4d} {h}x{w}  rel err {err:.2e}  least-squares scale {ratio:.6f}  S={float(S):.3g}")

# BatchNorm-backward inputs of the first up-sampling block, recomputed on the host from the dumped tensors
name = [st[1] for st in plan if st[0] == "conv" and (st[6] & E.CONV_UPSAMPLE)][-1]
buf, S, (b, h, w, C) = dbg[name]
z, k2f, k3f, dbeta, dgamma, sums, S_in = dbg[name + ":bn"]
def unsplit(t, C):
    o = t.view(torch.float16).float().reshape(-1, 2 * C).cpu().double()
    return o[:, :C] + o[:, C:]
gg, zz = unsplit(buf, C), unsplit(z, C)
mask = (k2f.cpu().double() * zz + k3f.cpu().double()) > 0
host_sdy = (gg * mask).sum(0)
print(name, "host sum dy vs kernel sums[:,0]:", float((host_sdy - sums[:, 0].cpu()).abs().max() / host_sdy.abs().max()))
print("kernel dbeta*S vs host:", float((dbeta.cpu().double() * float(S_in) - host_sdy).abs().max() / host_sdy.abs().max()))
# reference: activation and bias gradient of that layer
refmod = [st for st in plan if st[1] == name][0]
print("ref dbeta vs host/S:", float((refmod[5].bias.grad - host_sdy / float(S_in)).abs().max() / refmod[5].bias.grad.abs().max()))
act = {}
hk = relu_of[name].register_forward_hook(lambda m_, i, o: act.__setitem__("a", o.detach()))
ref.encode(m.double(), s.double(), g.double())
hk.remove()
a_ref = act["a"].permute(0, 2, 3, 1).reshape(-1, C)
a_ours = torch.relu(k2f.cpu().double() * zz + k3f.cpu().double())
print("activation err:", float((a_ours - a_ref).abs().max()), "mask mismatches:", int(((a_ref > 0) != mask).sum()), "of", mask.numel())
