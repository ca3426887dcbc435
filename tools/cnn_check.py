"""CNN_embedding (two Conv2d(3, stride 2) + ReLU + Linear) at the config-5 size against torch's own conv2d on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from asr_amd.Models.Transformer_encoder import CNN_embedding
dev = torch.device("cuda:0")
B = int(os.environ.get("CNN_B", "32")); T = int(os.environ.get("CNN_T", "1500"))
torch.manual_seed(0)
m = CNN_embedding(80, 256, 256).to(dev)
x = torch.randn(B, T, 80, device=dev)
w = None
DT = torch.float64 if os.environ.get("CNN_F64") else torch.float32
P64 = {k: p.detach().to(DT).requires_grad_(True) for k, p in m.named_parameters()}
def ref(x):
    y = x.to(DT).unsqueeze(1)
    y = torch.relu(torch.nn.functional.conv2d(y, P64["conv.0.weight"], P64["conv.0.bias"], stride=2))
    y = torch.relu(torch.nn.functional.conv2d(y, P64["conv.2.weight"], P64["conv.2.bias"], stride=2))
    b, c, t, f = y.shape
    return torch.nn.functional.linear(y.transpose(1, 2).contiguous().view(b, t, c * f), P64["out.0.weight"])
y0, _ = m(x, None)
wgt = torch.randn_like(y0)
(y0 * wgt).sum().backward()
g0 = {k: p.grad.clone() for k, p in m.named_parameters()}
m.zero_grad()
y1 = ref(x)
(y1 * wgt.to(DT)).sum().backward()
print("reference dtype", DT, "forward max rel err", float((y0 - y1).abs().max() / y1.abs().max()))
for k, p in P64.items():
    print(k, "grad rel err", float((g0[k] - p.grad).abs().max() / p.grad.abs().max()))
