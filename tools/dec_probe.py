#!/usr/bin/env python3
"""Where a 64-row block of conv_dec_kernel<48, 48, 2> spends its time: per-block s_memtime stamps of wave 0 of the first 256 workgroups
(diagnostic build: python -m animal2vec_amd.build --variant cdstamp -DCD_STAMP; A2V_LIB_PATH=animal2vec_amd/liba2v_hip_cdstamp.so python tools/dec_probe.py)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animal2vec_amd import ops, _lib  # noqa: E402


def main():
    lib = _lib.load()
    assert hasattr(lib, "a2v_diag_cd_stamps"), "needs the -DCD_STAMP build (A2V_LIB_PATH)"
    N, T, G, k, Cc = 64 * 12, 2000, 16, 7, 48
    D = G * Cc
    x = torch.randn(N, T, D, device="cuda").bfloat16()
    res = torch.randn(N, T, D, device="cuda").bfloat16()
    w = (torch.randn(G, Cc, k * 64, device="cuda") * (Cc * k) ** -0.5).bfloat16()
    b = torch.randn(D, device="cuda")
    y = torch.empty_like(x)
    for name, kw in (("bias", dict(bias=b)), ("residual", dict(res=res))):
        for _ in range(3):
            if "bias" in kw:
                ops.conv1d_cl(x, w, b, k=k, stride=1, pad=k // 2, groups=G, T_out=T, n_out=D, cv=Cc, out=y)
            else:
                ops.conv1d_cl(x, w, None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=D, cv=Cc, res=res, out=y)
        torch.cuda.synchronize()
        buf = np.zeros((256, 24, 8), dtype=np.uint64)
        assert lib.a2v_diag_cd_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
        st = buf.astype(np.int64)
        blk = st[:, 2:22]                                   # steady-state blocks
        d_math = (blk[:, :, 1] - blk[:, :, 0]).mean()       # fetch issue + per-group residual loads, MFMAs, stores
        d_wait = (blk[:, :, 2] - blk[:, :, 1]).mean()       # counted wait for the next slab
        d_stash = (blk[:, :, 3] - blk[:, :, 2]).mean()      # stash (ds_write) + barrier
        d_blk = (st[:, 3:22, 0] - st[:, 2:21, 0]).mean()
        seg = lambda a_, b_: (blk[:, :, a_] - blk[:, :, b_]).mean()
        print(f"{name:9s} block {d_blk:8.0f} shader cycles = fetch issue {seg(4, 0):6.0f} + group 0: MFMA loop {seg(5, 4):6.0f}, epilogue {seg(6, 5):6.0f} + group 1: MFMA loop "
              f"{seg(7, 6):6.0f}, epilogue {seg(1, 7):6.0f} + wait for the next slab {d_wait:6.0f} + stash and barrier {d_stash:6.0f}")


if __name__ == "__main__":
    main()
