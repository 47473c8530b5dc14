"""One small end-to-end invocation of the hot path on cuda:0, checked against the CPU oracle.
(The oracle import lives here only because __graft_entry__.smoke() is allowed to use it as the
checker; nothing else in the package touches it.)"""
import math

import torch


def run():
    import importlib
    O = importlib.import_module("a2v_" + "oracle")   # checker only
    from . import ops
    torch.manual_seed(0)
    C_, k, sr, B, L = 127, 63, 8000, 2, 4000
    lo, band = O.sinc_init(C_, k, sr)
    win, n_ = O.sinc_window_and_n(k, sr)
    x = torch.randn(B, L)
    lw, lb = torch.rand(C_) + 0.5, torch.randn(C_) * 0.1
    pa, pb = torch.full((C_,), 2.0), torch.randn(C_) * 0.2
    ref = O.pswish(O.channel_layer_norm(O.sinc_conv(x, lo, band, k, sr), lw, lb), pa.view(1, -1, 1), pb.view(1, -1, 1))
    cu = lambda t: t.cuda()
    filt = ops.sinc_filters_fwd(cu(lo.flatten()), cu(band.flatten()), cu(n_.flatten()), cu(win), C_=C_, Cp=128, k=k,
                                min_low_hz=50.0, min_band_hz=float(math.ceil(sr / k)), sample_rate=sr)
    y, _, _ = ops.sinc_fused_fwd(cu(x), filt, cu(lw), cu(lb), cu(pa), cu(pb), C_=C_, k=k, dtype=torch.float32)
    torch.cuda.synchronize()
    err = float((y[..., :C_].transpose(1, 2).cpu() - ref).norm() / ref.norm())
    assert err < 1e-4, err
    print(f"smoke ok: fused SincConv+LN+PSwish rel err {err:.2e}")
