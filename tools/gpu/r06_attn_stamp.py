import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from animal2vec_amd import ops
H, hd, N, S = 16, 64, 768, 152
slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device="cuda")
sc = torch.ones(H, device="cuda")
g = torch.Generator().manual_seed(1)
pos = torch.stack([torch.randperm(2000, generator=g)[:S].sort()[0] for _ in range(N)]).int().cuda()
qkv = torch.randn(N, S, 3 * H * hd, device="cuda").bfloat16()
ops.debug_set(40, 16)
for _ in range(3):
    o, lse = ops.attention_fwd(qkv, pos, slopes, sc, H, drop=(0.1, 1234))
torch.cuda.synchronize()
st = lse[1].view(N, H, S)[:, :, :32].reshape(N * H, 4, 8)[:, :, :5].float().cpu()
names = ["prologue (DMA + barrier)", "S^T loop item 0", "softmax item 0", "PV item 0", "whole wave"]
for w in range(4):
    print("wave", w, " ".join(f"{names[k]}: {st[:, w, k].median().item():.0f}" for k in range(5)))
print("median over all waves:", [round(st[:, :, k].median().item()) for k in range(5)])
