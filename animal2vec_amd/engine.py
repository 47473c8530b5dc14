"""Manual forward / backward schedule of the animal2vec pretraining step on one MI355X.

Python only sequences kernel launches on the current HIP stream (static shapes, no host syncs, so a
whole step is graph-capturable); every FLOP and byte is moved by the kernels behind include/a2v.h.
The schedule mirrors Data2VecMultiModel.forward (reference nn/data2vec2.py:516-991):

  student : SincConv+LN+PSwish -> conv stack -> project -> clone/mask -> pos-conv -> gather kept tokens
            -> prenet + blocks (ALiBi from kept positions) -> decoder -> masked MSE vs teacher targets
  teacher : (no grad, EMA weights) pos-conv -> prenet + blocks on all T frames -> instance-norm targets

and the backward is written out by hand in exactly the reverse order, which is also the order of the
flat gradient buffer (params.py), so finished gradient slices can be all-reduced while the rest of
backward still runs (`on_grad_ready` callback).
"""
from __future__ import annotations

import os

from typing import Callable, Dict, List, Optional

import torch

from . import ops
from .config import Data2VecMultiConfig
from .params import P, Packs, ParamStore

_DACT = os.environ.get("A2V_NO_DACT") is None      # (A/B switch for the gelu'-in-the-forward-epilogue flavour)


class Engine:
    def __init__(self, cfg: Data2VecMultiConfig, store: ParamStore, dtype=torch.bfloat16, x3: bool = False, teacher_half=None):
        """dtype: the activation storage (bf16 = throughput mode, fp32 = parity mode).  x3 (with fp32): the bf16x3 parity mode --
        every plain Linear product (forward, input gradient, weight gradient) runs as three exact bf16 products on the bf16
        matrix cores over split operands (ops.split_bf16, include/a2v.h: A2V_BF16X3); everything else is the fp32 mode."""
        self.tstream = None      # (first: __del__ -> close() must find it even when the constructor raises below)
        self._ws_keys = []
        self.cfg, self.store, self.dtype = cfg, store, dtype
        self.x3 = bool(x3)
        self.k_cap = {}          # (sequences, frames) -> kept-frame capacity of the student's allocations (ops.set_row_cap)
        dev = store.device
        # the EMA teacher's forward runs on a side stream, concurrently with the student's forward (A2V_TEACHER_STREAM=0: off)
        self.tstream = torch.cuda.Stream(dev) if (os.environ.get("A2V_TEACHER_STREAM", "1") == "1" and dev.type == "cuda") else None
        # caller-owned scratch (include/a2v.h, ABI 10), before the first library call: one workspace for the stream the engine is built
        # on and one for the teacher's side stream; the library keeps nothing of its own for them (another stream a caller switches to
        # later falls back to the library's static sets)
        if dev.type == "cuda":
            with torch.cuda.device(dev):
                self._ws_keys.append(ops.bind_workspace())
                if self.tstream is not None:
                    self._ws_keys.append(ops.bind_workspace(self.tstream))
        self.sp = Packs(store, dtype, teacher=False, need_grad=True, x3=self.x3)
        # The EMA teacher in IEEE half (bf16 mode only; OPT-IN: A2V_TEACHER_F16=1 or teacher_half=True).  Its no-grad forward
        # (data2vec2.py:755-844) produces the regression targets; half keeps 11 significand bits where bf16 keeps 8 at the same matrix
        # rate, and the reference's own recipe is fp16 (a2v_large_pretrain_best.yaml:4).  It runs in liba2v_hip_f16.so
        # (ops.half_library) on fp16 operand images.  Measured (profiles/r06_f16_teacher.txt): target error at full size
        # 4.4e-2 -> 1.9e-2 (the floor is now the bf16 front end's 1.1e-2 on the features the teacher is fed), on weights 200 updates from
        # init 2.0e-2 -> 5e-3; but the step is 1.1 % slower (the fp16 MFMAs hold a lower clock at the power cap) and the student's
        # GRADIENTS move away from the oracle's near a minimum (median 0.9e-2 -> 1.5e-2, worst 3.8e-2 -> 9.3e-2): with both networks in
        # bf16 the rounding of prediction and target is largely common-mode and cancels in their difference.  Hence not the default.
        want_half = teacher_half if teacher_half is not None else os.environ.get("A2V_TEACHER_F16", "0") == "1"
        self.teacher_half = bool(want_half and dtype == torch.bfloat16 and not self.x3 and dev.type == "cuda" and ops.have_half_library())
        self.tp = Packs(store, torch.float16 if self.teacher_half else dtype, teacher=True, need_grad=False, x3=self.x3)
        self.sp.refresh()
        self.refresh_teacher_packs()
        self.layers = cfg.conv_layers
        self.sv: Dict[str, object] = {}
        self.on_grad_ready: Optional[Callable[[str], None]] = None
        self.stats = torch.zeros(4 * cfg.embed_dim + 2, dtype=torch.float64, device=dev)
        c0 = self.layers[0][0]
        k0 = self.layers[0][1]
        self.dfilt = torch.zeros(c0, k0, dtype=torch.float32, device=dev)
        self.sinc_cp = ops.sinc_out_channels(c0, k0)      # channel pitch of the SincConv output (128, or 320 for the blocked kernels)
        self.dpar = torch.zeros(4, self.sinc_cp, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ helpers
    def close(self):
        """Hands the caller-owned workspaces back (drains the streams first: a persistent GEMM queued there may still use the tile
        counters); a side stream that picked up one of the library's static sets before it was bound returns it too."""
        ts, self.tstream = getattr(self, "tstream", None), None
        keys, self._ws_keys = getattr(self, "_ws_keys", []), []
        try:
            if ts is not None:
                ts.synchronize()
            for k in keys:
                ops.release_workspace(k)
            if ts is not None:
                from . import _lib
                _lib.load().a2v_stream_release(ts.cuda_stream)
        except Exception:
            pass

    def __del__(self):
        self.close()

    def refresh_student_packs(self):
        self.sp.refresh()

    def refresh_teacher_packs(self):
        if self.teacher_half:
            with ops.half_library():
                self.tp.refresh()
        else:
            self.tp.refresh()

    def _pads(self, i):
        """(pad_left) of feature-extractor layer i: 'same' (left = (k-1)//2) for stride 1 else ceil(s/2)."""
        _, k, s = self.layers[i]
        return (k - 1) // 2 if s == 1 else (s + 1) // 2

    # ------------------------------------------------------------------ front end
    def frontend_fwd(self, source: torch.Tensor, save: bool):
        """source (B,L) fp32 -> feats (B,T,D) [project_features output].  reference base.py:194-213"""
        cfg, st, sp, sv = self.cfg, self.store, self.sp, self.sv
        p = st.p
        pre = P + "local_encoder.conv_layers."
        B, L = source.shape
        c0, k0, _ = self.layers[0]
        filt = ops.sinc_filters_fwd(p[pre + "0.0.low_hz_"], p[pre + "0.0.band_hz_"], st.sinc_n, st.sinc_window, C_=c0,
                                    Cp=self.sinc_cp, k=k0, min_low_hz=50.0, min_band_hz=st.sinc_min_band,
                                    sample_rate=cfg.audio.sample_rate)
        y, mean, rstd = ops.sinc_fused_fwd(source, filt, p[pre + "0.2.1.weight"], p[pre + "0.2.1.bias"],
                                           p[pre + "0.3.p_swish_alpha"], p[pre + "0.3.p_swish_beta"], C_=c0, k=k0,
                                           dtype=self.dtype, save_stats=save)
        if save:
            sv["src"], sv["filt"], sv["y0"], sv["st0"] = source, filt, y, (mean, rstd)
        T_in = L
        for i in range(1, len(self.layers)):
            dim, k, s = self.layers[i]
            pl = self._pads(i)
            T_out = T_in if s == 1 else (T_in + 2 * pl - k) // s + 1
            c = ops.conv1d_cl(y, sp.w[pre + f"{i}.0.weight"], None, k=k, stride=s, pad=pl, groups=1, T_out=T_out, n_out=dim)
            yn, mean, rstd = ops.layernorm_fwd(c, p[pre + f"{i}.2.1.weight"], p[pre + f"{i}.2.1.bias"], 1e-5, act=1,
                                               save_stats=save)
            if save:
                sv[f"c{i}"], sv[f"st{i}"], sv[f"y{i}"] = c, (mean, rstd), yn
            y, T_in = yn, T_out
        ln, mean, rstd = ops.layernorm_fwd(y, p[P + "project_features.1.weight"], p[P + "project_features.1.bias"], 1e-5,
                                           save_stats=save)
        D = cfg.embed_dim
        feats = ops.linear(ln.view(B * T_in, -1), sp.w[P + "project_features.2.weight"], p[P + "project_features.2.bias"])
        if save:
            sv["pf_ln"], sv["pf_st"] = ln, (mean, rstd)
        return feats.view(B, T_in, D)

    def frontend_bwd(self, dfeat: torch.Tensor, project_only: bool = False):
        """project_only: stop after project_features (finetune runs the conv feature extractor under no_grad,
        local_grad_mult = 0, base.py:196-206)."""
        cfg, st, sp, sv = self.cfg, self.store, self.sp, self.sv
        p, g = st.p, st.g
        pre = P + "local_encoder.conv_layers."
        B, T, D = dfeat.shape
        ln = sv["pf_ln"]
        cf = ln.shape[-1]
        d2 = dfeat.view(B * T, D)
        ops.linear_wgrad(ln.view(B * T, cf), d2, g[P + "project_features.2.weight"])
        ops.colsum(d2, g[P + "project_features.2.bias"])
        dln = ops.linear(d2, sp.wt[P + "project_features.2.weight"])
        nl = len(self.layers)
        dy = ops.layernorm_bwd(dln.view(B, T, cf), sv[f"y{nl - 1}"], p[P + "project_features.1.weight"],
                               p[P + "project_features.1.bias"], *sv["pf_st"], dgamma=g[P + "project_features.1.weight"],
                               dbeta=g[P + "project_features.1.bias"])
        self._ready("project")
        if project_only:
            return
        for i in reversed(range(1, nl)):
            dim, k, s = self.layers[i]
            pl = self._pads(i)
            wname = pre + f"{i}.0.weight"
            m = sp.meta[wname]
            x_in = sv[f"y{i - 1}"]
            T_in, T_out = x_in.shape[1], dy.shape[1]
            dc = ops.layernorm_bwd(dy, sv[f"c{i}"], p[pre + f"{i}.2.1.weight"], p[pre + f"{i}.2.1.bias"], *sv[f"st{i}"],
                                   act=1, dgamma=g[pre + f"{i}.2.1.weight"], dbeta=g[pre + f"{i}.2.1.bias"])
            gw = sp.gw[wname]                      # cleared by the previous unpack_wgrad (zero at allocation)
            ops.conv1d_cl_wgrad(x_in, dc, gw, k=k, stride=s, pad=pl, groups=1, T_out=T_out, cv=m["Cgp"] if i == 1 else m["Cg"])
            ops.unpack_wgrad(gw, g[wname], GN=dim, Cg=m["Cg"], k=k, Cgp=m["Cgp"])
            if s > 1:
                dA = ops.linear(dc.view(B * T_out, dim), sp.wt[wname])
                if i == 1:
                    cp = self.sinc_cp
                    dy = torch.zeros(B, T_in, cp, dtype=self.dtype, device=dfeat.device) if m["Cgp"] != cp else \
                        torch.empty(B, T_in, cp, dtype=self.dtype, device=dfeat.device)
                    ops.col2im(dA, B=B, T_in=T_in, T_out=T_out, C_=m["Cgp"], k=k, s=s, p=pl, out=dy)
                else:
                    dy = ops.col2im(dA, B=B, T_in=T_in, T_out=T_out, C_=m["Cgp"], k=k, s=s, p=pl)
            else:
                dy = ops.conv1d_cl(dc, sp.wt[wname], None, k=k, stride=1, pad=k - 1 - pl, groups=1, T_out=T_in,
                                   n_out=m["Cg"], cv=m["N"])
        c0, k0, _ = self.layers[0]
        self.dfilt.zero_()
        self.dpar.zero_()
        ops.sinc_fused_bwd(sv["src"], sv["filt"], p[pre + "0.2.1.weight"], p[pre + "0.2.1.bias"],
                           p[pre + "0.3.p_swish_alpha"], p[pre + "0.3.p_swish_beta"], dy, *sv["st0"], self.dfilt, self.dpar,
                           C_=c0, k=k0)
        g[pre + "0.2.1.weight"].add_(self.dpar[0, :c0])
        g[pre + "0.2.1.bias"].add_(self.dpar[1, :c0])
        g[pre + "0.3.p_swish_alpha"].view(-1).add_(self.dpar[2, :c0])
        g[pre + "0.3.p_swish_beta"].view(-1).add_(self.dpar[3, :c0])
        ops.sinc_filters_bwd(p[pre + "0.0.low_hz_"], p[pre + "0.0.band_hz_"], st.sinc_n, st.sinc_window, self.dfilt, C_=c0,
                             k=k0, min_low_hz=50.0, min_band_hz=st.sinc_min_band, sample_rate=cfg.audio.sample_rate,
                             dlow=g[pre + "0.0.low_hz_"], dband=g[pre + "0.0.band_hz_"])
        self._ready("frontend")

    # ------------------------------------------------------------------ transformer block
    def _site(self, base, site):
        """per-site dropout seed (None when dropout is off)"""
        return None if base is None else (base * 1000003 + site * 7919 + 12345) & 0xFFFFFFFF

    def block_fwd(self, x, prefix, pk: Packs, params, pos, alibi_scale, H, save: Optional[str], want_t=False,
                  drop_base=None, site=0, dp=None):
        """AltBlock post-LN with ffn targets (reference modules.py:328-337).  x (N,S,D).
        dp: dropout probabilities (att, proj, mlp, act) overriding the pretraining config's -- finetuning sets them
        from its own config (wav2vec2.py:94-131); act = the Mlp's two internal dropouts (timm Mlp drop1 / drop2)."""
        N, S, D = x.shape
        eps = self.cfg.norm_eps
        x2 = x.view(N * S, D)
        # (long sequences in frame order -- the teacher's 2000 frames, finetuning: the qkv product's epilogue also leaves the row-norm
        #  maxima the attention forward's ALiBi band limit is made of, ops.linear_qknorm)
        qkn = None
        if pos is None and S >= 512 and x.dtype != torch.float32:
            qkv, qkn = ops.linear_qknorm(x2, pk.w[prefix + "attn.qkv.weight"], params[prefix + "attn.qkv.bias"], S)
        else:
            qkv = ops.linear(x2, pk.w[prefix + "attn.qkv.weight"], params[prefix + "attn.qkv.bias"])
        c = self.cfg
        p_att, p_proj, p_mlp, p_act = dp if dp is not None else (c.attention_dropout, c.encoder_dropout, c.post_mlp_drop, 0.0)

        def mk(pv, k, salt=0):
            if drop_base is None or pv <= 0:
                return None
            return (pv, (self._site(drop_base, 4 * site + k) + salt) & 0xFFFFFFFF)

        d_att, d_proj, d_mlp = mk(p_att, 0), mk(p_proj, 1), mk(p_mlp, 2)
        d_act1, d_act2 = mk(p_act, 3), mk(p_act, 3, 0x632BE5AB)
        att, lse = ops.attention_fwd(qkv.view(N, S, 3 * D), pos, self.store.alibi_slopes, alibi_scale, H,
                                     need_lse=save is not None, drop=d_att, qk_norms=qkn)
        att2 = att.view(N * S, D)        # (the object kept for the backward: in the bf16x3 mode its split image rides on it)
        s1 = ops.linear(att2, pk.w[prefix + "attn.proj.weight"], params[prefix + "attn.proj.bias"], res=x2, drop=d_proj)
        x1, m1, r1 = ops.layernorm_fwd(s1, params[prefix + "norm1.weight"], params[prefix + "norm1.bias"], eps,
                                       save_stats=save is not None)
        hid = pk.w[prefix + "mlp.fc1.weight"].shape[0]
        # (bf16 mode: gelu' is kept in 8 bits, A2V_EPI_DACT8 -- a quarter of fc1's output bytes; hid % 8 == 0 is the store's alignment)
        d8 = _DACT and ops.DACT8 and x.dtype == torch.bfloat16 and hid % 8 == 0 and d_act1 is None
        hpre = ops._E(N * S, hid, dtype=torch.uint8 if d8 else x.dtype, device=x.device) if save is not None else None
        # (dact: what is kept for the backward is gelu'(fc1 out), formed with gelu in one evaluation -- not the pre-activation)
        # (bf16x3: nothing but fc2's two products reads the hidden activation -- it leaves the epilogue as its split image, ops.X3Image)
        gact = ops.linear(x1, pk.w[prefix + "mlp.fc1.weight"], params[prefix + "mlp.fc1.bias"], gelu=True, pre_out=hpre,
                          drop=d_act1, dact=_DACT, out_image=self.x3)
        t = None
        s2 = None
        if want_t or d_act2 is not None:
            # the FFN output t (after the Mlp's own output dropout) is a result of its own (layer_results): keep it
            # apart from the residual sum x1 + post_mlp_dropout(t)
            t = ops.linear(gact, pk.w[prefix + "mlp.fc2.weight"], params[prefix + "mlp.fc2.bias"], drop=d_act2)
            if d_mlp is not None:
                s2 = ops.add_drop(x1, t, drop=d_mlp)
                x2o, m2, r2 = ops.layernorm_fwd(s2, params[prefix + "norm2.weight"], params[prefix + "norm2.bias"], eps,
                                                save_stats=save is not None)
            else:
                s2 = ops._EL(t) if save is not None else None
                x2o, m2, r2 = ops.layernorm_fwd(t, params[prefix + "norm2.weight"], params[prefix + "norm2.bias"], eps,
                                                res=x1, save_stats=save is not None, sum_out=s2)
        else:
            s2 = ops.linear(gact, pk.w[prefix + "mlp.fc2.weight"], params[prefix + "mlp.fc2.bias"], res=x1, drop=d_mlp)
            x2o, m2, r2 = ops.layernorm_fwd(s2, params[prefix + "norm2.weight"], params[prefix + "norm2.bias"], eps,
                                            save_stats=save is not None)
        if save is not None:
            self.sv[save] = dict(x=x2, qkv=qkv, att=att, att2=att2, lse=lse, s1=s1, st1=(m1, r1), x1=x1, hpre=hpre, g=gact, s2=s2,
                                 st2=(m2, r2), N=N, S=S, drops=(d_att, d_proj, d_mlp), act=(d_act1, d_act2))
        return x2o.view(N, S, D), (t.view(N, S, D) if t is not None else None)

    def block_bwd(self, dx, prefix, pos, alibi_scale, H, key, dt=None):
        """dt: gradient arriving directly at the block's FFN output t (finetune: the head reads the mean of the last
        average_top_k_layers layer results)."""
        sv = self.sv.pop(key)
        st, sp = self.store, self.sp
        p, g = st.p, st.g
        N, S = sv["N"], sv["S"]
        D = dx.shape[-1]
        dx = dx.reshape(N * S, D)
        d_att, d_proj, d_mlp = sv["drops"]
        d_act1, d_act2 = sv["act"]
        # ds2: gradient of (x1 + drop(mlp)) -> residual path; ds2d: the same through the dropout mask -> mlp branch
        ds2, ds2d = ops.layernorm_bwd(dx, sv["s2"], p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], *sv["st2"],
                                      dgamma=g[prefix + "norm2.weight"], dbeta=g[prefix + "norm2.bias"],
                                      dxsum=g[prefix + "mlp.fc2.bias"] if d_act2 is None else None, drop=d_mlp, drop_mode=2)
        dt2 = dt.reshape(N * S, D) if dt is not None else None
        if d_act2 is not None:                 # back through the Mlp's output dropout: everything that arrives at t
            ds2d = ops.add_drop(None, ds2d, dt2, drop=d_act2)
            ops.colsum(ds2d, g[prefix + "mlp.fc2.bias"])
        elif dt2 is not None:
            ds2d = ops.mean_layers([ds2d, dt2], scale=1.0)
            ops.colsum(dt2, g[prefix + "mlp.fc2.bias"])          # dxsum above only saw the LN2 path
        # (bf16x3: dh is read by fc1's input-gradient and weight-gradient products only: it leaves the epilogue as its split image)
        dh = ops.linear(ds2d, sp.wt[prefix + "mlp.fc2.weight"], gelu_bwd_aux=sv["hpre"], colsum=g[prefix + "mlp.fc1.bias"],
                        drop=d_act1, dact=_DACT, out_image=self.x3 and d_act1 is None and ops.X3_DH_IMAGE)
        ops.linear_wgrad_pair(sv["g"], ds2d, g[prefix + "mlp.fc2.weight"], sv["x1"], dh, g[prefix + "mlp.fc1.weight"])
        dx1 = ops.linear(dh, sp.wt[prefix + "mlp.fc1.weight"], res=ds2)
        ds1, ds1d = ops.layernorm_bwd(dx1, sv["s1"], p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], *sv["st1"],
                                      dgamma=g[prefix + "norm1.weight"], dbeta=g[prefix + "norm1.bias"],
                                      dxsum=g[prefix + "attn.proj.bias"], drop=d_proj, drop_mode=2)
        datt = ops.linear(ds1d, sp.wt[prefix + "attn.proj.weight"])
        dqkv = ops.attention_bwd(sv["qkv"].view(N, S, 3 * D), sv["att"], datt.view(N, S, D), sv["lse"], pos,
                                 st.alibi_slopes, alibi_scale, H, g[P + "alibi_scale"].view(-1), drop=d_att,
                                 bias_grad=g[prefix + "attn.qkv.bias"])
        dq2 = dqkv.view(N * S, 3 * D)
        ops.linear_wgrad_pair(sv["att2"], ds1d, g[prefix + "attn.proj.weight"],
                              sv["x"], dq2, g[prefix + "attn.qkv.weight"])
        dxin = ops.linear(dq2, sp.wt[prefix + "attn.qkv.weight"], res=ds1)
        return dxin.view(N, S, D)

    # ------------------------------------------------------------------ grouped conv + LN + GELU stacks
    def posconv_fwd(self, feats, pk: Packs, params, mask_u8, cb, save: bool, rows=None, post=None):
        """relative_positional_encoder (audio.py:90-113) on x.repeat_interleave(cb)*(1-mask) (student) or
        on feats (teacher); the clone/mask view is taken by the A-operand loader, never materialised.
        rows (int32, flat n*T + t of the kept frames): the reference gathers the kept frames from the (N,T,D)
        output (base.py:300-314), so the LAST layer (conv, LayerNorm, GELU) is evaluated at those rows only and the
        result is compact (len(rows), D); `post` (compact) is added to it in the LayerNorm kernel."""
        cfg = self.cfg
        a = cfg.audio
        B, T, D = feats.shape
        k = cfg.pos_kernel
        N = B * cb
        x = feats
        depth = a.conv_pos_depth
        for i in range(1, depth + 1):
            w = P + f"relative_positional_encoder.{i}.0.weight"
            first = i == 1 and mask_u8 is not None
            ri = rows if i == depth else None
            bias_i = params[P + f"relative_positional_encoder.{i}.0.bias"]
            if first and ri is None and ops.conv_sparse_ok(x, pk.w[w], k=k, groups=a.conv_pos_groups, n_out=D):
                # the masked clones are zero at 92 % of their frames: products per kept row, scatter-added into the output window
                c = ops.conv1d_cl_sparse_in(x, pk.w[w], bias_i, k=k, pad=k // 2, groups=a.conv_pos_groups, T=T, n_out=D, n_seq=N,
                                            rowmask=mask_u8, bdiv=cb)
            else:
                c = ops.conv1d_cl(x, pk.w[w], bias_i, k=k, stride=1, pad=k // 2,
                                  groups=a.conv_pos_groups, T_out=T, n_out=D, rowmask=mask_u8 if first else None,
                                  bdiv=cb if first else 1, n_batch=N if first else None, cv=pk.meta[w]["Cg"], rowidx=ri)
            y, mean, rstd = ops.layernorm_fwd(c, None, None, 1e-5, act=1, save_stats=save,
                                              post=post if ri is not None else None)
            if save:
                self.sv[f"pc{i}"] = dict(x=x, c=c, st=(mean, rstd))
            x = y
        return x

    def posconv_bwd(self, dtok, mask_u8, cb, rows, ids_restore, T):
        """dtok (N,K,D): gradient of the kept tokens = gradient of the compact last-layer output.  Returns the
        gradient wrt the (masked, cloned) pos-conv input AT THE KEPT ROWS, with dtok already added (N,K,D): the
        masked rows of that input are constants (zeros), so the first layer's input gradient is only formed there."""
        cfg, st, sp = self.cfg, self.store, self.sp
        a = cfg.audio
        g = st.g
        k = cfg.pos_kernel
        G = a.conv_pos_groups
        N, K, D = dtok.shape
        depth = a.conv_pos_depth
        dy = dtok.view(N * K, D)
        for i in reversed(range(1, depth + 1)):
            sv = self.sv.pop(f"pc{i}")
            w = P + f"relative_positional_encoder.{i}.0.weight"
            m = sp.meta[w]
            first, last = i == 1, i == depth
            dc = ops.layernorm_bwd(dy, sv["c"], None, None, *sv["st"], act=1,
                                   dxsum=g[P + f"relative_positional_encoder.{i}.0.bias"])
            gw = sp.gw[w]                          # cleared by the previous unpack_wgrad (zero at allocation)
            ops.conv1d_cl_wgrad(sv["x"], dc, gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T,
                                rowmask=mask_u8 if first else None, bdiv=cb if first else 1, n_batch=N if first else None,
                                cv=m["Cg"], rowidx=rows if last else None, rows_kept=last)
            ops.unpack_wgrad(gw, g[w], GN=D, Cg=m["Cg"], k=k, Cgp=m["Cgp"])
            if last and not first and ops.conv_sparse_ok(dc, sp.wt[w], k=k, groups=G, n_out=D, rows_per_seq=K):
                # dc only exists at the kept frames: its products are scatter-added into the dense input gradient (no zero-filled copy)
                dy = ops.conv1d_cl_sparse_in(dc.view(N * K, D), sp.wt[w], None, k=k, pad=k - 1 - k // 2, groups=G, T=T, n_out=D, n_seq=N,
                                             rowidx=rows)
                continue
            if last:                                   # the input gradient needs dc as a dense (N,T,D) operand
                dc = ops.scatter_tokens(dc.view(N, K, D), ids_restore, T)
            if first:
                return ops.conv1d_cl(dc, sp.wt[w], None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=D,
                                     cv=m["N"], rowidx=rows, res=dtok.view(N * K, D)).view(N, K, D)
            dy = ops.conv1d_cl(dc, sp.wt[w], None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=D,
                               cv=m["N"])

    def posconv_bwd_dense(self, dy):
        """All T frames (finetune: no masking, no clones).  dy (B,T,D) -> gradient wrt the pos-conv input."""
        cfg, st, sp = self.cfg, self.store, self.sp
        a = cfg.audio
        g = st.g
        k = cfg.pos_kernel
        G = a.conv_pos_groups
        N, T, D = dy.shape
        for i in reversed(range(1, a.conv_pos_depth + 1)):
            sv = self.sv.pop(f"pc{i}")
            w = P + f"relative_positional_encoder.{i}.0.weight"
            m = sp.meta[w]
            dc = ops.layernorm_bwd(dy, sv["c"], None, None, *sv["st"], act=1,
                                   dxsum=g[P + f"relative_positional_encoder.{i}.0.bias"])
            gw = sp.gw[w]                          # cleared by the previous unpack_wgrad (zero at allocation)
            ops.conv1d_cl_wgrad(sv["x"], dc, gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T, cv=m["Cg"])
            ops.unpack_wgrad(gw, g[w], GN=D, Cg=m["Cg"], k=k, Cgp=m["Cgp"])
            dy = ops.conv1d_cl(dc, sp.wt[w], None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=D,
                               cv=m["N"])
        return dy

    # ------------------------------------------------------------------ finetune path
    def finetune_forward(self, source, head, *, save=True, top_k=None, aug=None):
        """Wav2VecEncoderModOut.forward (wav2vec2.py:430-482): extract_features, mean of the last
        average_top_k_layers layer results, Linear head.  source (B,L) fp32 -> padded logits (B*T, Cp) in the
        activation dtype; with save=True everything finetune_backward() needs is kept.
        aug=None: eval form (mask=False, no layerdrop, no dropout).  aug (dict, train form; finetune.py builds it):
          mask_t (B,T) uint8 / mask_c (B,D) uint8  -- apply_mask, base.py:457-485: frames replaced by N(0, noise_std)
                                                      (or zero), then channels zeroed; the mask is applied BEFORE the
                                                      pos-conv, which therefore sees the masked input (base.py:268-298)
          noise_std, noise_seed (device hash noise) or replacement ((num_masked, D) explicit rows, for reproducibility)
          active: list of bool, prenet blocks then main blocks -- layerdrop outcome (modules.py:89-94, data2vec2.py:651-655)
          p_in / p_final                                   -- dropout_input (before the main blocks), final_dropout (before the head)
          drop_seed (int or None), p_drop / p_act / p_att  -- cfg.dropout (prenet, proj, post-mlp), activation_dropout
                                                      (the Mlp's two dropouts), attention_dropout (wav2vec2.py:94-128)"""
        cfg = self.cfg
        D = cfg.embed_dim
        top_k = cfg.average_top_k_layers if top_k is None else top_k
        x, ts, ctx = self.encode_features(source, save=save, aug=aug)
        B, T = ctx["B"], ctx["T"]
        d_fin = ctx.pop("d_fin_spec")
        sel = ts[-top_k:]
        assert sel, "layerdrop removed every block"
        xm = ops.mean_layers([t for _, t in sel]).view(B * T, D)
        if d_fin is not None:
            xm = ops.add_drop(None, xm, drop=d_fin)                             # final_dropout (wav2vec2.py:467)
        logits = ops.linear(xm, head.w_op, head.b_op)                           # (B*T, Cp), Cp = classes padded to 16
        if save:
            self.sv["ft"] = dict(ctx, xm=xm, sel=[i for i, _ in sel], d_fin=d_fin)
        return logits

    def encode_features(self, source, *, save=False, aug=None):
        """Data2VecMultiModel.forward(features_only=True, mask=aug has masks) (data2vec2.py:629-728): student weights, all T
        frames (remove_masked False, clone_batch 1).  Returns (x (B,T,D), [(block index, FFN output t)], ctx) where ctx
        holds what finetune_backward() needs besides the per-block saves.  `aug` as in finetune_forward."""
        cfg, st, sp = self.cfg, self.store, self.sp
        p = st.p
        H, D = cfg.num_heads, cfg.embed_dim
        a = cfg.audio
        ops.X3_WGRAD = self.x3
        feats = self.frontend_fwd(source, save=save)
        B, T, _ = feats.shape
        asc = p[P + "alibi_scale"].view(-1)
        db = dp = d_pre = None
        n_blocks = a.prenet_depth + cfg.depth
        active = [True] * n_blocks
        masks = None
        if aug is not None:
            if aug.get("mask_t") is not None or aug.get("mask_c") is not None:
                masks = (aug.get("mask_t"), aug.get("mask_c"))
                if aug.get("replacement") is not None:      # reproducibility hook: explicit (num_masked, D) noise rows
                    feats = feats.clone()
                    feats[masks[0].bool()] = aug["replacement"].to(feats.dtype)
                    feats = ops.mask_frames(feats, None, masks[1])
                else:
                    feats = ops.mask_frames(feats, masks[0], masks[1], noise_std=aug.get("noise_std", 0.0),
                                            seed=aug.get("noise_seed", 0))
            if aug.get("active") is not None:
                active = list(aug["active"])
                assert len(active) == n_blocks
            db = aug.get("drop_seed")
            if db is not None:
                pd = aug.get("p_drop", 0.0)           # p_proj / p_mlp / p_pre: per-site overrides of cfg.dropout
                dp = (aug.get("p_att", 0.0), aug.get("p_proj", pd), aug.get("p_mlp", pd), aug.get("p_act", 0.0))
                if aug.get("p_pre", pd) > 0:
                    d_pre = (aug.get("p_pre", pd), self._site(db, 9000))
        x_pos = self.posconv_fwd(feats, sp, p, None, 1, save=save)
        tok = torch.empty_like(feats) if save else None
        x, m0, r0 = ops.layernorm_fwd(feats, p[P + "context_encoder.norm.weight"], p[P + "context_encoder.norm.bias"],
                                      cfg.norm_eps, res=x_pos, save_stats=save, sum_out=tok, drop=d_pre)
        for i in range(a.prenet_depth):
            if active[i]:
                x, _ = self.block_fwd(x, P + f"context_encoder.blocks.{i}.", sp, p, None, asc, H, save=f"fpb{i}" if save else None,
                                      drop_base=db, site=i, dp=dp)
        d_in = d_fin = None
        if aug is not None and db is not None:
            if aug.get("p_in", 0.0) > 0:
                d_in = (aug["p_in"], self._site(db, 9002))
                x = ops.add_drop(None, x, drop=d_in)          # dropout_input (data2vec2.py:646-647)
            if aug.get("p_final", 0.0) > 0:
                d_fin = (aug["p_final"], self._site(db, 9003))
        ts = []
        for i in range(cfg.depth):
            if active[a.prenet_depth + i]:
                x, t = self.block_fwd(x, f"blocks.{i}.", sp, p, None, asc, H, save=f"fmb{i}" if save else None, want_t=True,
                                      drop_base=db, site=a.prenet_depth + i, dp=dp)
                ts.append((i, t))
        ctx = dict(tok=tok, st=(m0, r0), B=B, T=T, active=active, masks=masks, d_pre=d_pre, d_in=d_in, d_fin_spec=d_fin)
        return x, ts, ctx

    def finetune_backward(self, dlog, head, *, head_only=False):
        """dlog (B*T, Cp): gradient of the loss wrt the padded logits (padded columns zero).  Accumulates encoder and
        project_features gradients into store.grad and the head's into head.gw / head.gb; the conv feature extractor is
        frozen (feature_grad_mult 0.0, finetune_mixup_100.yaml:85 -> local_grad_mult 0, base.py:196-206).
        head_only: the first freeze_finetune_updates updates train the classifier alone (wav2vec2.py:440-445)."""
        cfg, st = self.cfg, self.store
        p, g = st.p, st.g
        H, D = cfg.num_heads, cfg.embed_dim
        a = cfg.audio
        ops.X3_WGRAD = self.x3
        c = self.sv.pop("ft")
        B, T = c["B"], c["T"]
        asc = p[P + "alibi_scale"].view(-1)
        # ---- head
        ops.linear_wgrad(c["xm"], dlog, head.gw)
        ops.colsum(dlog, head.gb)
        if head_only:
            self.sv.clear()
            return
        dxm = ops.linear(dlog, head.wt_op)                                       # (B*T, D)
        if c.get("d_fin") is not None:
            dxm = ops.add_drop(None, dxm, drop=c["d_fin"])
        sel = c["sel"]
        dt = ops.mean_layers([dxm], scale=1.0 / len(sel)).view(B, T, D)
        # the last executed block's output x feeds nothing; every selected block receives dt at its FFN output
        self.features_backward(c, None, {i: dt for i in sel})

    def features_backward(self, c, dx, dts):
        """Backward of encode_features(save=True): dx (B,T,D) or None = gradient arriving at the last block's output x, dts {block
        index: (B,T,D)} = gradients arriving at the blocks' FFN outputs (layer_results, data2vec2.py:722-728).  Accumulates the
        encoder's and project_features' gradients into store.grad; the conv feature extractor is frozen as in finetune_backward.
        `c` is the context encode_features returned (with its per-block saves still in self.sv)."""
        cfg, st = self.cfg, self.store
        p, g = st.p, st.g
        H, D = cfg.num_heads, cfg.embed_dim
        a = cfg.audio
        ops.X3_WGRAD = self.x3
        B, T = c["B"], c["T"]
        asc = p[P + "alibi_scale"].view(-1)
        active = c["active"]
        dev = st.grad.device
        dxe = torch.zeros(B, T, D, dtype=self.dtype, device=dev) if dx is None else dx.to(self.dtype).contiguous()
        for i in reversed(range(cfg.depth)):
            if active[a.prenet_depth + i]:
                dxe = self.block_bwd(dxe, f"blocks.{i}.", None, asc, H, f"fmb{i}", dt=dts.get(i))
            self._ready(f"block{i}")
        if c.get("d_in") is not None:
            dxe = ops.add_drop(None, dxe.contiguous(), drop=c["d_in"])
        for i in reversed(range(a.prenet_depth)):
            if active[i]:
                dxe = self.block_bwd(dxe, P + f"context_encoder.blocks.{i}.", None, asc, H, f"fpb{i}")
            self._ready(f"prenet{i}")
        dtok = ops.layernorm_bwd(dxe, c["tok"], p[P + "context_encoder.norm.weight"], p[P + "context_encoder.norm.bias"],
                                 *c["st"], dgamma=g[P + "context_encoder.norm.weight"], dbeta=g[P + "context_encoder.norm.bias"],
                                 drop=c["d_pre"], drop_mode=1)
        dpos_in = self.posconv_bwd_dense(dtok)
        self._ready("posconv")
        dfeat = ops.mean_layers([dtok, dpos_in], scale=1.0)                      # x = feats_masked + pos_conv(feats_masked)
        if c["masks"] is not None:
            dfeat = ops.mask_frames(dfeat, c["masks"][0], c["masks"][1], backward=True, out=dfeat)
        self.frontend_bwd(dfeat, project_only=True)
        self._ready("frontend")
        self.sv.clear()

    def finetune_forward_backward(self, source, target, head, *, backward=True, gscale=1.0, top_k=None, aug=None,
                                  head_only=False):
        """finetune_forward + the summed sigmoid focal loss of FinetuneCrossEntropyCriterion (criterions.py:243-248,
        accumulated into head.loss_sum) + finetune_backward.  target (B,T,C) fp32 multi-hot (soft after target
        mixup).  Returns logits (B,T,C)."""
        logits = self.finetune_forward(source, head, save=backward, top_k=top_k, aug=aug)
        C_ = head.n_labels
        M = logits.shape[0]
        tgt = target.reshape(M, C_).contiguous()
        dlog = ops.focal_loss(logits, tgt, head.loss_sum, alpha=head.alpha, gamma=head.gamma, gscale=gscale, want_grad=backward)
        B = source.shape[0]
        out = logits.view(B, M // B, -1)[..., :C_]
        if backward:
            self.finetune_backward(dlog, head, head_only=head_only)
        return out

    def decoder_fwd(self, dec_in, save: bool):
        """Decoder1d (modules.py:137-192): grouped conv + LN(no affine) + GELU (+ residual) x L, then proj."""
        cfg, st, sp = self.cfg, self.store, self.sp
        dec = cfg.audio.decoder
        p = st.p
        N, T, D = dec_in.shape
        x = dec_in
        k = dec.decoder_kernel
        for i in range(dec.decoder_layers):
            w = P + f"decoder.blocks.{i}.0.weight"
            c = ops.conv1d_cl(x, sp.w[w], p[P + f"decoder.blocks.{i}.0.bias"], k=k, stride=1, pad=k // 2,
                              groups=dec.decoder_groups, T_out=T, n_out=dec.decoder_dim, cv=sp.meta[w]["Cg"])
            resid = x if (dec.decoder_residual and x.shape[-1] == dec.decoder_dim) else None
            y, mean, rstd = ops.layernorm_fwd(c, None, None, 1e-5, act=1, save_stats=save, post=resid)
            if save:
                self.sv[f"dc{i}"] = dict(x=x, c=c, st=(mean, rstd), resid=resid is not None)
            x = y
        out = ops.linear(x.view(N * T, dec.decoder_dim), sp.w[P + "decoder.proj.weight"], p[P + "decoder.proj.bias"])
        if save:
            self.sv["dec_last"] = x
        return out.view(N, T, D)

    def decoder_bwd(self, dout):
        cfg, st, sp = self.cfg, self.store, self.sp
        dec = cfg.audio.decoder
        g = st.g
        N, T, D = dout.shape
        Dd = dec.decoder_dim
        k = dec.decoder_kernel
        G = dec.decoder_groups
        d2 = dout.view(N * T, D)
        x_last = self.sv.pop("dec_last")
        ops.linear_wgrad(x_last.view(N * T, Dd), d2, g[P + "decoder.proj.weight"])
        ops.colsum(d2, g[P + "decoder.proj.bias"])
        dy = ops.linear(d2, sp.wt[P + "decoder.proj.weight"]).view(N, T, Dd)
        for i in reversed(range(dec.decoder_layers)):
            sv = self.sv.pop(f"dc{i}")
            w = P + f"decoder.blocks.{i}.0.weight"
            m = sp.meta[w]
            dc = ops.layernorm_bwd(dy, sv["c"], None, None, *sv["st"], act=1, dxsum=g[P + f"decoder.blocks.{i}.0.bias"])
            gw = sp.gw[w]                          # cleared by the previous unpack_wgrad (zero at allocation)
            ops.conv1d_cl_wgrad(sv["x"], dc, gw, k=k, stride=1, pad=k // 2, groups=G, T_out=T, cv=m["Cg"])
            ops.unpack_wgrad(gw, g[w], GN=Dd, Cg=m["Cg"], k=k, Cgp=m["Cgp"])
            n_in = sv["x"].shape[-1]
            dy = ops.conv1d_cl(dc, sp.wt[w], None, k=k, stride=1, pad=k - 1 - k // 2, groups=G, T_out=T, n_out=n_in,
                               res=dy if sv["resid"] else None, cv=m["N"])
        self._ready("decoder")
        return dy        # (N,T,D) gradient wrt decoder input

    def _ready(self, what: str):
        if self.on_grad_ready is not None:
            self.on_grad_ready(what)

    # ------------------------------------------------------------------ teacher
    @torch.no_grad()
    def teacher_targets(self, feats):
        """Teacher section + make_targets (data2vec2.py:779-846,1023-1066): feats (B,T,D) -> y fp32."""
        if self.teacher_half:
            with ops.half_library():
                return self._teacher_targets(feats.to(torch.float16))
        return self._teacher_targets(feats)

    def _teacher_targets(self, feats):
        cfg, st, tp = self.cfg, self.store, self.tp
        t = st.t
        B, T, D = feats.shape
        H = cfg.num_heads
        asc = t[P + "alibi_scale"].view(-1)
        x_pos = self.posconv_fwd(feats, tp, t, None, 1, save=False)
        x, _, _ = ops.layernorm_fwd(feats, t[P + "context_encoder.norm.weight"], t[P + "context_encoder.norm.bias"],
                                    cfg.norm_eps, res=x_pos, save_stats=False)
        for i in range(cfg.audio.prenet_depth):
            x, _ = self.block_fwd(x, P + f"context_encoder.blocks.{i}.", tp, t, None, asc, H, save=None)
        ts: List[torch.Tensor] = []
        first_target = cfg.depth - cfg.average_top_k_layers
        for i in range(cfg.depth):
            x, tt = self.block_fwd(x, f"blocks.{i}.", tp, t, None, asc, H, save=None, want_t=i >= first_target)
            if tt is not None:
                ts.append(tt)
        return ops.instnorm_targets(ts, eps=1e-5)

    # ------------------------------------------------------------------ whole step
    def forward(self, source, mask_u8, ids_keep, ids_restore, noise, *, save=True, keep_debug=False, dropout_seed=None):
        """Forward of one micro-batch (teacher targets + student + decoder + loss).  source (B,L) fp32;
        mask_u8 (B*cb,T) 1=masked; ids_keep (B*cb,K) int32 (ascending kept frames); ids_restore (B*cb,T)
        int32; noise (B*cb,T-K,D) fp32 (row j <-> j-th masked frame) or a (std, seed) pair (drawn in-kernel).  Adds loss/variance sums to
        self.stats; with save=True keeps what backward() needs.  dropout_seed (int) switches the student's
        train-mode dropouts on (prenet 0.1, attention 0.1, proj 0.1, post-mlp 0.1, decoder input 0.1: yaml:139-147,
        data2vec2.py:74-77); the teacher always runs in eval mode (data2vec2.py:779-780)."""
        cfg, st, sp = self.cfg, self.store, self.sp
        p = st.p
        cb, H, D = cfg.clone_batch, cfg.num_heads, cfg.embed_dim
        a = cfg.audio
        dbg = {}
        ops.X3_WGRAD = self.x3
        ops.X3_BACKWARD = False
        ops.X3_KEEP = False                  # (set for the student's part below; a forward that raised must not leave it on)
        N, K = ids_keep.shape
        # student activations at the capacity K_cap this shape was primed for (trainer.prime_memory), whatever K the masks drew
        ops.set_row_cap(N, K, self.k_cap.get((N, mask_u8.shape[1])))
        feats = self.frontend_fwd(source, save=save)                           # (B,T,D)
        B, T, _ = feats.shape
        asc = p[P + "alibi_scale"].view(-1)
        # ---- teacher targets from the student's local features (data2vec2.py:811-818): no gradient, independent of the
        # student until the loss -> on a side stream, so its kernels fill the CUs the student's kernels leave idle
        # (single-round GEMMs occupy 240 of 256 CUs, every kernel has a ramp and a tail) and vice versa
        if self.tstream is not None:
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
            with torch.cuda.stream(self.tstream):
                self.tstream.wait_event(fork)
                y = self.teacher_targets(feats)     # (the library keeps one scratch set per stream: include/a2v.h)
                join = torch.cuda.Event()
                join.record(self.tstream)
            y.record_stream(main)
        else:
            join = None
            y = self.teacher_targets(feats)
        # ---- student context encoder (base.py:215-344)
        # kept frames as flat rows n*T + t: the last pos-conv layer is only evaluated there (compact), with the
        # gathered local features added in its LayerNorm kernel
        rows = (torch.arange(N, device=ids_keep.device, dtype=torch.int32).unsqueeze(1) * T + ids_keep).reshape(-1)
        ops.X3_KEEP = self.x3 and save          # bf16x3: the split images of the saved activations serve the backward's weight gradients
        tok_f = ops.gather_tokens(feats, ids_keep, cb=cb)                      # (N,K,D)
        tok = self.posconv_fwd(feats, sp, p, mask_u8, cb, save=save, rows=rows, post=tok_f.view(N * K, D)).view(N, K, D)
        x_pos = tok
        db = dropout_seed
        d_pre = (a.prenet_dropout, self._site(db, 9000)) if db is not None and a.prenet_dropout > 0 else None
        d_dec = (a.decoder.input_dropout, self._site(db, 9001)) if db is not None and a.decoder.input_dropout > 0 else None
        x, m0, r0 = ops.layernorm_fwd(tok, p[P + "context_encoder.norm.weight"], p[P + "context_encoder.norm.bias"],
                                      cfg.norm_eps, save_stats=save, drop=d_pre)
        if save:
            self.sv["tok"], self.sv["tok_st"] = tok, (m0, r0)
        for i in range(a.prenet_depth):
            x, _ = self.block_fwd(x, P + f"context_encoder.blocks.{i}.", sp, p, ids_keep, asc, H,
                                  save=f"pb{i}" if save else None, drop_base=db, site=i)
        if keep_debug:
            dbg["prenet_out"] = x
        d_in = (cfg.dropout_input, self._site(db, 9002)) if db is not None and cfg.dropout_input > 0 else None
        if d_in is not None:                               # data2vec2.py:646-647
            x = ops.add_drop(None, x, drop=d_in)
        for i in range(cfg.depth):
            x, _ = self.block_fwd(x, f"blocks.{i}.", sp, p, ids_keep, asc, H, save=f"mb{i}" if save else None,
                                  drop_base=db, site=a.prenet_depth + i)
        if keep_debug:
            dbg["encoder_out"] = x
        # ---- decoder + loss
        dec_in = ops.decoder_input(x, noise, ids_restore, drop=d_dec)          # (N,T,D)
        dec_out = self.decoder_fwd(dec_in, save=save)
        ops.X3_KEEP = False
        loss_rows = torch.empty(N, T, D, dtype=torch.float32, device=x.device) if keep_debug else None
        if join is not None:
            torch.cuda.current_stream().wait_event(join)
        dx = ops.d2v_loss(dec_out, y, mask_u8, cb, stats=self.stats, want_dx=save, gscale=1.0, loss_rows=loss_rows)
        if keep_debug:
            dbg.update(local_features=feats, targets=y, x_pos=x_pos, decoder_in=dec_in, decoder_out=dec_out,
                       loss_rows=loss_rows)
        if not save:
            ops.set_row_cap(0, 0, None)
        if save:
            self.sv["ctx"] = dict(dx=dx, mask_u8=mask_u8, ids_keep=ids_keep, ids_restore=ids_restore, T=T, d_pre=d_pre,
                                  d_dec=d_dec, rows=rows, d_in=d_in, row_cap=(N, K, self.k_cap.get((N, T))))
        return dbg

    def backward(self, gscale: float = 1.0):
        """Backward of the micro-batch last run through forward(save=True); accumulates into store.grad."""
        cfg, st = self.cfg, self.store
        p, g = st.p, st.g
        cb, H = cfg.clone_batch, cfg.num_heads
        a = cfg.audio
        ops.X3_WGRAD = self.x3
        ops.X3_BACKWARD = self.x3
        c = self.sv.pop("ctx")
        ops.set_row_cap(*c["row_cap"])
        dx, mask_u8, ids_keep, ids_restore, T = c["dx"], c["mask_u8"], c["ids_keep"], c["ids_restore"], c["T"]
        if gscale != 1.0:
            dx.mul_(gscale)
        asc = p[P + "alibi_scale"].view(-1)
        ddec_in = self.decoder_bwd(dx)                                         # (N,T,D)
        dxe = ops.gather_tokens(ddec_in, ids_keep, cb=1, drop=c["d_dec"])      # (N,K,D)
        for i in reversed(range(cfg.depth)):
            dxe = self.block_bwd(dxe, f"blocks.{i}.", ids_keep, asc, H, f"mb{i}")
            self._ready(f"block{i}")
        if c.get("d_in") is not None:
            dxe = ops.add_drop(None, dxe.contiguous(), drop=c["d_in"])
        for i in reversed(range(a.prenet_depth)):
            dxe = self.block_bwd(dxe, P + f"context_encoder.blocks.{i}.", ids_keep, asc, H, f"pb{i}")
            self._ready(f"prenet{i}")
        dtok = ops.layernorm_bwd(dxe, self.sv.pop("tok"), p[P + "context_encoder.norm.weight"],
                                 p[P + "context_encoder.norm.bias"], *self.sv.pop("tok_st"),
                                 dgamma=g[P + "context_encoder.norm.weight"], dbeta=g[P + "context_encoder.norm.bias"],
                                 drop=c["d_pre"], drop_mode=1)
        dxk = self.posconv_bwd(dtok, mask_u8, cb, c["rows"], ids_restore, T)   # (N,K,D): pos-conv path + dtok, kept rows
        self._ready("posconv")
        dfeat = ops.clone_reduce(None, dxk, ids_restore, cb)                   # (B,T,D)
        ops.set_row_cap(0, 0, None)
        self.frontend_bwd(dfeat)
        ops.X3_BACKWARD = False
        self.sv.clear()

    def forward_backward(self, source, mask_u8, ids_keep, ids_restore, noise, *, backward=True, gscale=1.0,
                         keep_debug=False, dropout_seed=None):
        """One micro-batch: forward then (optionally) backward.  Accumulates parameter gradients into
        store.grad and loss/variance sums into self.stats."""
        dbg = self.forward(source, mask_u8, ids_keep, ids_restore, noise, save=backward, keep_debug=keep_debug,
                           dropout_seed=dropout_seed)
        if backward:
            self.backward(gscale)
        return dbg

    # ------------------------------------------------------------------ features only (no masking)
    @torch.no_grad()
    def extract_features(self, source):
        """Data2VecMultiModel.forward(features_only=True, mask=False) (data2vec2.py:629-728, 1112-1123):
        student weights, all T frames, no token removal.  Returns (x, layer_results)."""
        x, ts, _ = self.encode_features(source, save=False, aug=None)
        return x, [t for _, t in ts]
