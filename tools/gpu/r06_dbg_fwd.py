import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
from animal2vec_amd import ops
import a2v_oracle as O
torch.manual_seed(0)
def ref(qkv, pos, slopes, sc, H):
    N, S, W = qkv.shape; hd = W // (3 * H)
    q, k, v = qkv.reshape(N, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    bias = O.alibi_bias_from_positions(pos.long(), H, sc.view(1, 1, H, 1, 1))
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1) + bias).softmax(-1)
    return (att @ v).transpose(1, 2).reshape(N, S, H * hd)
import sys as _s
if len(_s.argv) > 1: ops.debug_set(40, int(_s.argv[1]))
for S in (64, 152):
    for N, H in ((1, 2), (3, 4)):
        qkv = torch.randn(N, S, 3 * H * 64).bfloat16().float()
        slopes = torch.tensor(O.alibi_slopes(H))
        pos = torch.arange(S).unsqueeze(0).expand(N, -1).int().contiguous()
        for scv in (1.3, 83.2):
            sc = torch.full((H,), scv)
            r = ref(qkv, pos, slopes, sc, H)
            for need in (True, False):
                out, _ = ops.attention_fwd(qkv.cuda().bfloat16(), None, slopes.cuda(), sc.cuda(), H, need_lse=need)
                d = (out.float().cpu() - r).abs()
                bad = (d > 0.05).nonzero()
                print(f"S={S} N={N} H={H} sc={scv} need_lse={need}: max err {d.max().item():.3g}, bad {len(bad)}", bad[:6].tolist())
