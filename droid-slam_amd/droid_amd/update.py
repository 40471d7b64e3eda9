"""ConvGRU update operator on MI355X: host-side mirror of the reference's UpdateModule
(droid_slam/droid_net.py:78-143), ConvGRU (droid_slam/modules/gru.py:5-33) and GraphAgg (droid_net.py:44-75).

Same constructor-free interface as the reference module object: ``UpdateModule.forward(net, inp, corr, flow, ii, jj)``
with the reference's tensor shapes, and ``load_state_dict`` accepts the reference's parameter names
(``update.*`` of a droid.pth checkpoint), so ``factor_graph.py`` can call it unchanged.

All convolutions and the GRU algebra run in the hand-written implicit-GEMM kernel of csrc/conv.hip
(droid_backends.conv2d_nhwc: fp16 MFMA, fp32 accumulate, fused bias/activation/gate epilogues); activations are
kept NHWC fp16 between layers, the 448-channel GRU input is never concatenated, z|r and the two head stems are
single launches.  torch is used only for allocation, the per-edge 128-vector "global context" GEMVs and the
scatter-mean over source frames (index_add_), i.e. plumbing.  There is no CPU path.
"""
import os

import torch
import torch.nn.functional as F

import droid_backends

from ._cache import tensor_cache_key

EPI_LINEAR, EPI_RELU, EPI_SIGMOID, EPI_GRU_ZR, EPI_GRU_Q, EPI_GLO, EPI_SOFTPLUS_001, EPI_HEADS, EPI_HEADS0 = range(9)
COR_PLANES = 4 * 49
COR_NHWC = 224           # channel-last correlation features: 4 levels x (49 + 7 zero channels)


def corr_channel_map():
    """index [224] into the reference's 196 correlation channels (level*49 + xoff*7 + yoff) for the channel-last
    layout written by droid_backends.corr_pyramid_lookup_nhwc ([level][...][yoff*7 + xoff], 56 per level);
    -1 = zero padding."""
    m = torch.full((COR_NHWC,), -1, dtype=torch.long)
    for l in range(4):
        for a in range(7):
            for b in range(7):
                m[l * 56 + b * 7 + a] = l * 49 + a * 7 + b
    return m

# reference parameter shapes (Cout, Cin, k); droid_net.py:83-108, gru.py:8-17, droid_net.py:47-57
PARAM_SHAPES = {
    "corr_encoder.0": (128, COR_PLANES, 1), "corr_encoder.2": (128, 128, 3),
    "flow_encoder.0": (128, 4, 7), "flow_encoder.2": (64, 128, 3),
    "weight.0": (128, 128, 3), "weight.2": (2, 128, 3),
    "delta.0": (128, 128, 3), "delta.2": (2, 128, 3),
    "gru.convz": (128, 448, 3), "gru.convr": (128, 448, 3), "gru.convq": (128, 448, 3),
    "gru.w": (128, 128, 1),
    "gru.convz_glo": (128, 128, 1), "gru.convr_glo": (128, 128, 1), "gru.convq_glo": (128, 128, 1),
    "agg.conv1": (128, 128, 3), "agg.conv2": (128, 128, 3),
    "agg.eta.0": (1, 128, 3), "agg.upmask.0": (576, 128, 1),
}


def empty_state_dict(dtype=torch.float32):
    """zero tensors under the reference's parameter names (shape template for random-init / checkpoint loading)"""
    sd = {}
    for k, (co, ci, ks) in PARAM_SHAPES.items():
        sd[k + ".weight"] = torch.zeros(co, ci, ks, ks, dtype=dtype)
        sd[k + ".bias"] = torch.zeros(co, dtype=dtype)
    return sd


def _round_up(v, m):
    return (v + m - 1) // m * m


def pack_conv(weight, bias, cin_pad=None):
    """[Cout,Cin,KH,KW] -> ([CoutPad, Kpad] f16 with k = (dy*KW+dx)*CinPad + c, bias [CoutPad] f32)."""
    cout, cin, kh, kw = weight.shape
    cin_pad = cin if cin_pad is None else cin_pad
    w = weight.float()
    if cin_pad != cin:
        w = F.pad(w, (0, 0, 0, 0, 0, cin_pad - cin))
    w = w.permute(0, 2, 3, 1).reshape(cout, kh * kw * cin_pad)
    cout_pad, kpad = _round_up(cout, 32), _round_up(w.shape[1], 64)
    wp = torch.zeros(cout_pad, kpad, dtype=torch.float16, device=weight.device)
    wp[:cout, :w.shape[1]] = w.half()
    bp = torch.zeros(cout_pad, dtype=torch.float32, device=weight.device)
    bp[:cout] = bias.float()
    return wp.contiguous(), bp.contiguous()


def _dma_layout(cout_pad, ctot):
    """layout rule of `weights_halo`, the same as csrc/conv.hip dma_layout()"""
    if droid_backends.get_option("conv_dma") != 1:
        return False
    return cout_pad % 128 == 0 and ctot % 64 == 0 and ctot >= 256


def _halo2_layout(cout_pad, ctot):
    """the same rule as csrc/conv.hip halo2_layout()"""
    if droid_backends.get_option("conv_halo2") == 0:
        return False
    return not _dma_layout(cout_pad, ctot) and cout_pad % 128 == 0 and ctot % 32 == 0


def pack_conv_halo(weight):
    """Second, kernel-ordered copy of a 3x3 weight [Cout,Ctot,3,3] (csrc/conv.hip), or None if the shape is not eligible.
      * DH_CONV_DMA=1 (opt-in experiment), CoutPad % 128 == 0, Ctot % 64 == 0 and Ctot >= 256: [CoutPad/128, Ctot/64, 9, 128, 8, 8] f16 for the LDS-DMA kernel -- one
        (tap, 64-channel chunk) slab is 128 cout rows of 128 bytes, the 16-byte slot s of row r stored at slot
        s ^ ((r >> 1) & 7) (the bank swizzle of the kernel's ds_read addresses);
      * CoutPad % 128 == 0 and Ctot % 32 == 0 (default; DH_CONV_HALO2=0 disables): [CoutPad/128, Ctot/32, 3, 3, 128, 4, 8] for
        conv3x3_halo2_kernel -- a (32-channel chunk, kernel row) group is 3 x 128 rows of 64 bytes, the 16-byte slot s of
        row r stored at slot s ^ ((r >> 2) & 3);
      * otherwise [CoutPad/BN, Ctot/CK, 9, BN, CK] for the halo-tile kernels (BN = 32 / 64 with 32-channel chunks for the
        small heads, else BN = 128 with 16-channel chunks)."""
    cout, ctot, kh, kw = weight.shape
    if kh != 3 or kw != 3 or ctot % 32:
        return None
    cp = _round_up(cout, 32)                # CoutPad of pack_conv
    if cp == 64 and droid_backends.get_option("conv_halo64") and droid_backends.get_option("conv_halo2"):
        cp = 128                            # conv3x3_halo64_kernel reads the halo2 layout of the layer padded to 128 couts (csrc/conv.hip halo64_layout)
    w = torch.zeros(cp, ctot, 9, dtype=torch.float32, device=weight.device)
    w[:cout] = weight.float().reshape(cout, ctot, 9)
    if _dma_layout(cp, ctot):
        w = w.reshape(cp // 128, 128, ctot // 64, 8, 8, 9).permute(0, 2, 5, 1, 3, 4)        # [T, chunk, tap, row, slot, 8]
        r = torch.arange(128, device=weight.device)
        src = torch.arange(8, device=weight.device)[None, :] ^ ((r[:, None] >> 1) & 7)        # stored slot s' holds slot s'^sw
        w = torch.gather(w, 4, src[None, None, None, :, :, None].expand(*w.shape[:3], 128, 8, 8))
        return w.half().contiguous()
    if _halo2_layout(cp, ctot):
        w = w.reshape(cp // 128, 128, ctot // 32, 4, 8, 3, 3).permute(0, 2, 5, 6, 1, 3, 4)   # [T, chunk, dy, dx, row, slot, 8]
        r = torch.arange(128, device=weight.device)
        src = torch.arange(4, device=weight.device)[None, :] ^ ((r[:, None] >> 2) & 3)        # stored slot s' holds slot s'^sw
        w = torch.gather(w, 5, src[None, None, None, None, :, :, None].expand(*w.shape[:4], 128, 4, 8))
        return w.half().contiguous()
    bn = cp if cp in (32, 64) else 128      # cout tile of the kernel variant that will take this convolution
    ck = 16 if bn == 128 else 32            # halo_ck() of csrc/conv.hip
    if cp % bn:
        return None
    w = w.reshape(cp // bn, bn, ctot // ck, ck, 9).permute(0, 2, 4, 1, 3)
    return w.half().contiguous()


LAYOUT_AUTO, LAYOUT_WINO = 0, 4          # DH_CONV_LAYOUT_* of include/droid_hip.h


def pack_corr0_fused(w0):
    """corr_encoder.0.weight [128,196] (input channel = level*49 + xoff*7 + yoff, corr.py:46-50) -> [13,128,16] fp16 in the
    K order of csrc/corr_pyramid.hip pyr_lookup_corr0_kernel: k-step level*3 + s holds the level's samples kk = 16s..16s+15
    with kk = yoff*7 + xoff (the order the interpolation produces them in); the four samples kk = 48 share k-step 12."""
    out = torch.zeros(13, 128, 16, dtype=torch.float32, device=w0.device)
    for l in range(4):
        for kk in range(48):
            xoff, yoff = kk % 7, kk // 7
            out[l * 3 + kk // 16, :, kk % 16] = w0[:, l * 49 + xoff * 7 + yoff]
        out[12, :, l] = w0[:, l * 49 + 48]
    return out.half().contiguous()


def pack_conv_wino(weight):
    """[Cout,Ctot,3,3] -> [CoutPad/128, Ctot/32, 3 dy, 4 positions, 128, 4 slots, 8] f16 for conv3x3_wino_kernel (PROTOTYPE:
    Winograd F(2,3) along x): per kernel row the three taps g0 g1 g2 become (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2), computed in
    fp32 and rounded once; rows / slots ordered like the halo2 layout (slot s of row r stored at s ^ ((r >> 2) & 3)).
    None if the shape is not eligible (Cout padded to a multiple of 128, Ctot % 32 == 0)."""
    cout, ctot, kh, kw = weight.shape
    cp = _round_up(cout, 32)
    if kh != 3 or kw != 3 or ctot % 32 or cp % 128:
        return None
    g = torch.zeros(cp, ctot, 3, 3, dtype=torch.float32, device=weight.device)
    g[:cout] = weight.float()
    u = torch.stack([g[..., 0], 0.5 * (g[..., 0] + g[..., 1] + g[..., 2]), 0.5 * (g[..., 0] - g[..., 1] + g[..., 2]), g[..., 2]], -1)   # [cp, ctot, dy, t]
    u = u.reshape(cp // 128, 128, ctot // 32, 4, 8, 3, 4).permute(0, 2, 5, 6, 1, 3, 4)        # [T, chunk, dy, t, row, slot, 8]
    r = torch.arange(128, device=weight.device)
    src = torch.arange(4, device=weight.device)[None, :] ^ ((r[:, None] >> 2) & 3)
    u = torch.gather(u, 5, src[None, None, None, None, :, :, None].expand(*u.shape[:4], 128, 4, 8))
    return u.half().contiguous()


def pack_conv_7x7_c4(weight):
    """[128,4,7,7] -> [128, 7 dy, 8 dx (7 + one zero tap), 4 ch] f16: the kernel-ordered copy for conv7x7_c4_kernel
    (csrc/conv.hip): a k-step of 16 is four x-adjacent taps of one kernel row."""
    cout, cin, kh, kw = weight.shape
    if (cout, cin, kh, kw) != (128, 4, 7, 7):
        return None
    w = torch.zeros(128, 7, 8, 4, dtype=torch.float32, device=weight.device)
    w[:, :, :7, :] = weight.float().permute(0, 2, 3, 1)
    return w.reshape(128, 224).half().contiguous()


def _halo_layout_options():
    """the options that decide the layout of `weights_halo` (csrc/conv.hip dma_layout / halo2_layout)"""
    return droid_backends.get_option("conv_dma"), droid_backends.get_option("conv_halo2"), droid_backends.get_option("conv_halo64")


def transposed_state_dict(sd):
    """parameters (reference names) of the update operator of the TRANSPOSED image: k x k kernels transposed, the 7 x 7 lookup window of
    corr_encoder.0's 196 input channels (level*49 + xoff*7 + yoff) transposed; everything else unchanged.  With T = swapping the two
    image axes (and the window axes of the correlation features): operator(sd)(x) == T(operator(transposed_state_dict(sd))(T(x)))
    (tests/test_transposed_cpu.py checks the identity on the oracle)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() == 4 and v.shape[-1] > 1:
            v = v.transpose(-1, -2)
        elif k.endswith("corr_encoder.0.weight"):
            v = v.reshape(128, 4, 7, 7).transpose(2, 3).reshape(128, COR_PLANES, 1, 1)
        out[k] = v.contiguous()
    return out


class _Conv:
    def __init__(self, wp, bp, k, cout, wh=None, layout=LAYOUT_AUTO):
        self.w, self.b, self.k, self.cout, self.wh, self.layout = wp, bp, k, cout, wh, layout
        # the kernel-ordered copy `wh` was laid out for the option values of this moment; csrc/conv.hip picks its kernel from
        # the option values at launch time -> remember them, re-check whenever any option changed (options_epoch)
        self._epoch = droid_backends.options_epoch()
        self._layout = _halo_layout_options()

    def _check_layout(self):
        epoch = droid_backends.options_epoch()
        if epoch != self._epoch:
            if self.wh is not None and self.layout == LAYOUT_AUTO and _halo_layout_options() != self._layout:
                raise RuntimeError("conv weights were packed for (conv_dma, conv_halo2, conv_halo64) = %s, the options now say %s: "
                                   "call load_state_dict again after changing them" % (self._layout, _halo_layout_options()))
            self._epoch = epoch

    def __call__(self, inputs, epi, out=None, out_stride=None, gterm=None, aux0=None, aux1=None, red=None,
                 cinit=None, cinit_idx=None, cinit_off=0, out_raw_f32=False, out_tiled=False, glo=None):
        """out_tiled (with out_raw_f32): the fp32 output in the accumulator-tile layout of csrc/conv.hip (ConvParams::cinit_stride < 0),
        returned as [N, h*w/256 pixel tiles, cout/128 cout tiles, 32768 floats per tile] (frame-major: slices of whole frames stay
        valid); a `cinit` tensor of that shape is read back in that layout.
        glo = (weight [128,128] f16, bias [128] f32, red [N,128] f32 zeroed), EPI_GRU_Q only: the next iteration's global-context
        reduction on the new hidden state, fused behind the gate (dh_conv2d_nhwc_f16_ex3)"""
        self._check_layout()
        x0 = inputs[0]
        if out is None and out_tiled:
            assert out_raw_f32 and (x0.shape[1] * x0.shape[2]) % 256 == 0 and self.cout % 128 == 0
            out = torch.empty(x0.shape[0], x0.shape[1] * x0.shape[2] // 256, self.cout // 128, 32768, dtype=torch.float32, device=x0.device)
        if out is None and epi not in (EPI_GLO, EPI_HEADS0):
            out = torch.empty(x0.shape[0], x0.shape[1], x0.shape[2], self.cout,
                              dtype=torch.float32 if out_raw_f32 else torch.float16, device=x0.device)
        stride = 0 if out is None else (self.cout if out_tiled else out.shape[-1] if out_stride is None else out_stride)
        cinit_tiled = cinit is not None and cinit.dim() == 4 and cinit.shape[-1] == 32768 and cinit.shape[1] * 256 == x0.shape[1] * x0.shape[2]
        droid_backends.conv2d_nhwc(list(inputs), self.w, self.wh, self.b, self.k, self.k, self.cout, epi, out, stride,
                                   gterm, aux0, aux1, red, cinit, cinit_idx, cinit_off, out_raw_f32, self.layout, out_tiled, cinit_tiled,
                                   *(glo if glo is not None else (None, None, None)))
        return out


class _Fork:
    """`with _Fork(side, main) as b: ...` runs the block on the side stream after everything already enqueued on the caller's stream;
    `b.join(t, ...)` makes the caller's stream wait for the block and tells the allocator that it reads the tensors.  side = None: the
    block runs inline on the caller's stream."""

    def __init__(self, side, main):
        self.side, self.main, self.done, self.ctx = side, main, None, None

    def __enter__(self):
        if self.side is not None:
            self.side.wait_event(self.main.record_event())
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            self.done = self.side.record_event()
            self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        if self.side is not None:
            self.main.wait_event(self.done)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.main)


class UpdateModule:
    """Weight-compatible, inference-only replacement of droid_net.UpdateModule."""

    def __init__(self, device="cuda", share_inp_by_source_frame=False, canvas=True):
        self.device = torch.device(device)
        self.params = None
        self.last_dw = None
        self.last_glo = None                   # forward_nhwc(glo_next=True): the new hidden state's global-context sums [E,128] f32
        # image sizes outside the production tiling of csrc/conv.hip (64-pixel rows, four-row tiles) run on a zero-padded
        # 64-column canvas through the SAME kernels (forward_nhwc); False = the generic implicit-GEMM loop on the image itself
        self.canvas = canvas
        # forward() (the reference's interface, per-edge `inp`): treat edges with equal ii as sharing their context features
        self.share_inp_by_source_frame = share_inp_by_source_frame
        self._sd, self._twin = None, None      # the loaded tensors, kept for transposed_twin()
        # DH_UPDATE_STREAMS=1 (opt-in): the independent branches of the operator (flow encoder / global context + context term next to
        # the correlation encoder; GraphAgg next to the heads) on two side streams, so that the HBM- and latency-bound kernels among
        # them run under the MFMA-bound ones.  Measured at C3 (profiles/r04_r_update_streams_ab.txt): 48.22 vs 48.39 ms per operator
        # call -- the step is bound by the socket's power, and a joule costs the same on any stream; off by default.
        self.streams = os.environ.get("DH_UPDATE_STREAMS", "0") == "1"
        self._side = None
        self._derived = {}                     # slot -> (key of the source tensor, derived tensor, source kept alive): see _cached

    # ---- parameters ----------------------------------------------------------------------------
    def load_state_dict(self, sd, prefix=""):
        """sd: reference names ('gru.convz.weight', ...), optionally prefixed (e.g. 'update.')."""
        g = lambda n: sd[prefix + n].to(self.device)
        self._sd, self._twin = {k + s_: sd[prefix + k + s_] for k in PARAM_SHAPES for s_ in (".weight", ".bias")}, None
        self._derived = {}                     # (context terms of the previous weights)
        P = {}
        conv = lambda name, cin_pad=None: _Conv(*pack_conv(g(name + ".weight"), g(name + ".bias"), cin_pad),
                                                PARAM_SHAPES[name][2], PARAM_SHAPES[name][0],
                                                pack_conv_halo(g(name + ".weight")) if cin_pad is None else None)
        cmap = corr_channel_map().to(self.device)
        w0 = g("corr_encoder.0.weight").float()
        w0p = torch.zeros(128, COR_NHWC, 1, 1, device=self.device)
        w0p[:, cmap >= 0] = w0[:, cmap[cmap >= 0]]
        P["corr0"] = _Conv(*pack_conv(w0p, g("corr_encoder.0.bias")), 1, 128)
        # the same layer on the reference-layout lookup output [E,196,h,w] (droid_backends.corr0_nchw): [cout][208] fp16
        w0n = torch.zeros(128, 208, device=self.device)
        w0n[:, :COR_PLANES] = w0.reshape(128, COR_PLANES)
        P["corr0_nchw"] = (w0n.half().contiguous(), g("corr_encoder.0.bias").float().contiguous())
        # the same layer inside the lookup kernel (droid_backends.corr_pyramid_lookup_corr0): [13 k-steps][128][16] fp16
        P["corr0_fused"] = (pack_corr0_fused(w0.reshape(128, COR_PLANES)), P["corr0_nchw"][1])
        self.cmap = cmap
        P["corr2"] = conv("corr_encoder.2")
        P["flow0"] = conv("flow_encoder.0", 8)
        P["flow0"].wh = pack_conv_7x7_c4(g("flow_encoder.0.weight"))
        P["flow2"] = conv("flow_encoder.2")
        P["gru_w"] = conv("gru.w")
        # z | r as one convolution with 256 outputs
        wz, wr = g("gru.convz.weight"), g("gru.convr.weight")
        P["zr"] = _Conv(*pack_conv(torch.cat([wz, wr], 0), torch.cat([g("gru.convz.bias"), g("gru.convr.bias")], 0)), 3, 256,
                        pack_conv_halo(torch.cat([wz, wr], 0)))
        P["q"] = conv("gru.convq")
        # The context features `inp` are the same for every edge of a source frame (reference factor_graph.py:135,299:
        # video.inps[ii]).  Split the 448 input channels of the gate convolutions into (net | c, f) = 320 per-edge channels
        # and the 128 channels of inp: the latter's contribution to z | r | q is ONE convolution per FRAME
        # (`ctx`: 128 -> 384, fp32, no bias) that the per-edge convolutions (`zr_e`, `q_e`) start their accumulators from.
        wq = g("gru.convq.weight")
        edge_ch = torch.cat([torch.arange(0, 128), torch.arange(256, 448)]).to(self.device)
        wzr_e = torch.cat([wz, wr], 0)[:, edge_ch].contiguous()
        # (DH_CONV_WINO=1 / set_option("conv_wino", 1) before loading: the two per-edge gate convolutions through the Winograd
        # F(2,3) prototype kernel instead of conv3x3_halo2_kernel -- A/B measurements)
        wino = droid_backends.get_option("conv_wino") == 1
        gate = lambda w: (pack_conv_wino(w), LAYOUT_WINO) if wino else (pack_conv_halo(w), LAYOUT_AUTO)
        P["zr_e"] = _Conv(*pack_conv(wzr_e, torch.cat([g("gru.convz.bias"), g("gru.convr.bias")], 0)), 3, 256, *gate(wzr_e))
        wq_e = wq[:, edge_ch].contiguous()
        P["q_e"] = _Conv(*pack_conv(wq_e, g("gru.convq.bias")), 3, 128, *gate(wq_e))
        wctx = torch.cat([wz, wr, wq], 0)[:, 128:256].contiguous()
        P["ctx"] = _Conv(*pack_conv(wctx, torch.zeros(384, device=self.device)), 3, 384, pack_conv_halo(wctx))
        for n in ("z", "r", "q"):
            P["glo_" + n] = (g("gru.conv%s_glo.weight" % n).float().reshape(128, 128), g("gru.conv%s_glo.bias" % n).float())
        # k-major weights of the global-context GEMVs (droid_backends.glo_gemv): z | r together, q on its own
        P["glo_zr_t"] = (torch.cat([P["glo_z"][0], P["glo_r"][0]], 0).t().contiguous(), torch.cat([P["glo_z"][1], P["glo_r"][1]]).contiguous())
        P["glo_q_t"] = (P["glo_q"][0].t().contiguous(), P["glo_q"][1].contiguous())
        # the two head stems (delta.0 | weight.0) share their input: one convolution with 256 outputs;
        # the two 2-channel heads become one block-diagonal convolution on those 256 channels
        wh0 = torch.cat([g("delta.0.weight"), g("weight.0.weight")], 0)
        P["heads0"] = _Conv(*pack_conv(wh0, torch.cat([g("delta.0.bias"), g("weight.0.bias")], 0)), 3, 256, pack_conv_halo(wh0))
        w2 = torch.zeros(4, 256, 3, 3, device=self.device)
        w2[0:2, 0:128] = g("delta.2.weight").float()
        w2[2:4, 128:256] = g("weight.2.weight").float()
        P["heads2"] = _Conv(*pack_conv(w2, torch.cat([g("delta.2.bias"), g("weight.2.bias")], 0)), 3, 4, pack_conv_halo(w2))
        # fused form (csrc/conv.hip EPI_HEADS0): second-layer weights as [cout tile][tap*4 + output (36 of 64)][128 channels]
        w2p = torch.zeros(2, 64, 128, device=self.device)
        w2p[:, :36] = w2.reshape(4, 2, 128, 9).permute(1, 3, 0, 2).reshape(2, 36, 128)
        P["heads2_fused"] = (w2p.half().contiguous(), torch.cat([g("delta.2.bias"), g("weight.2.bias")], 0).float().contiguous())
        P["agg1"] = conv("agg.conv1")
        P["agg2"] = conv("agg.conv2")
        P["eta"] = conv("agg.eta.0")
        # round 6: the eta head (3x3, 128 -> 1) as the fused second layer of agg.conv2's launch (csrc/conv.hip EPI_HEADS0 with an `out`
        # pointer): weights in the heads' layout [cout tile][tap*4 + output (output 0 only)][128 channels]
        we = torch.zeros(1, 64, 128, device=self.device)
        we[0, 0:36:4] = g("agg.eta.0.weight").float().reshape(128, 9).t()
        be = torch.zeros(4, device=self.device); be[0] = g("agg.eta.0.bias").float()[0]
        P["eta_fused"] = (we.half().contiguous(), be.contiguous())
        P["upmask"] = conv("agg.upmask.0")
        self.params = P
        return self

    # ---- layout helpers (module boundary only) -----------------------------------------------------
    @staticmethod
    def to_nhwc(x, cpad=None):
        """[E,C,h,w] -> [E,h,w,Cpad] f16 contiguous."""
        x = x.permute(0, 2, 3, 1)
        if cpad is not None and cpad != x.shape[-1]:
            x = F.pad(x, (0, cpad - x.shape[-1]))
        return x.to(torch.float16).contiguous()

    # ---- the operator ----------------------------------------------------------------------------
    def context_term(self, inp_frames, tiled=None):
        """inp_frames [K,h,w,128] f16 -> [K,h,w,384] f32: the context features' share of the z | r | q pre-activations
        (no bias).  Depends only on the frames' context features, i.e. it can be kept for as long as the keyframes live;
        forward_nhwc recomputes it on every call unless the caller passes it in.
        Round 5: by default (option cinit_tiled) the 1536 bytes per pixel are stored in the ACCUMULATOR-TILE layout of the gate
        kernel (the registers of its 64 x 64 wave tiles, dumped by the producing launch and restored by the gate launches with
        16-byte loads: profiles/r05_f_conv_phase_timeline.txt shows the pixel-major form holding a workgroup 18 us before its first
        MFMA).  Same values; the tensor then has the shape [K, h*w/256, 3, 32768] and is only meaningful as `cinit=` of the gate convolutions."""
        P = self.params
        if tiled is None:
            # (the same conditions as csrc/conv.hip halo2_ok<EPI_GRU_ZR / Q> + staged_epilogue_ok: a gate launch that cannot take the
            # production kernel reads pixel-major start values only)
            tiled = (droid_backends.get_option("cinit_tiled") == 1 and droid_backends.get_option("conv_halo") == 1
                     and droid_backends.get_option("conv_halo2") == 1 and droid_backends.get_option("conv_epi_staged") == 1
                     and P["zr_e"].layout == LAYOUT_AUTO and P["ctx"].wh is not None
                     and inp_frames.shape[2] == 64 and inp_frames.shape[1] % 4 == 0)
        return P["ctx"]([inp_frames], EPI_LINEAR, out_raw_f32=True, out_tiled=bool(tiled))

    def corr0_layer(self, corr):
        """corr_encoder.0 (1x1, 196 -> 128, ReLU; droid_net.py:96-100) on correlation features [E,196,h,w] (reference layout; pixel
        counts that are a multiple of 128 go through droid_backends.corr0_nchw, others through the channel-last form) or on the
        level-planar channel-last [4,E,h,w,56] -> [E,h,w,128] f16"""
        P = self.params
        if corr.dim() == 4 and corr.shape[1] == COR_PLANES:
            if self.wants_reference_layout_corr(corr.shape[2], corr.shape[3]):
                return droid_backends.corr0_nchw(corr.contiguous(), P["corr0_nchw"][0], P["corr0_nchw"][1])
            corr = self.corr_to_nhwc(corr)
        return P["corr0"]([corr[0], corr[1], corr[2], corr[3]], EPI_RELU)

    @staticmethod
    def wants_reference_layout_corr(h, w):
        """True where forward_nhwc is fastest on the UNPADDED reference-layout correlation features [E,196,h,w]
        (CorrBlock.__call__ / corr_pyramid_lookup: 392 instead of 448 bytes per pixel leave the lookup): its first layer
        then transposes its tiles itself (droid_backends.corr0_nchw, 128-pixel tiles)"""
        return (h * w) % 128 == 0

    def segments(self, ii):
        """edges grouped by source frame for GraphAgg's mean (droid_net.py:66-67): (order [E], seg_off [K+1]).  Depends on the
        edge list only, so it is kept until `ii` changes (torch.unique has to synchronise to size its result)."""
        key = tensor_cache_key(ii)
        if key is None or getattr(self, "_seg_key", None) != key:
            _, ix, cnt = torch.unique(ii, return_inverse=True, return_counts=True)
            order = torch.argsort(ix, stable=True)
            seg_off = torch.zeros(cnt.numel() + 1, dtype=torch.int64, device=ii.device)
            seg_off[1:] = torch.cumsum(cnt, 0)
            self._seg_key, self._seg = key, (order, seg_off, ii)       # (ii kept alive: its address is part of the key)
        return self._seg[0], self._seg[1]

    def fuses_next_glo(self, h, w):
        """True where forward_nhwc(glo_next=True) computes the next iteration's global-context reduction inside the q gate's launch
        (the production 3x3 kernel on a 64-column image with staged epilogues; not on canvases, whose padding is re-zeroed after the gate)"""
        P = self.params
        return bool(w == 64 and h % 4 == 0 and droid_backends.get_option("conv_halo") == 1 and droid_backends.get_option("conv_halo2") == 1
                    and droid_backends.get_option("conv_epi_staged") == 1 and droid_backends.get_option("glo_fused") == 1
                    and P["q_e"].layout == LAYOUT_AUTO and P["q_e"].wh is not None and P["q"].wh is not None)

    def forward_nhwc(self, net, inp, corr, flow, ii, inp_frames=None, inp_index=None, ctx=None, corr0=None, _mask=None,
                     glo_red=None, glo_next=False, _glo=None, want_upmask=True):
        """net [E,h,w,128] f16 (updated IN PLACE), corr = [E,196,h,w] f16 in the reference's layout (pixel counts that are a
        multiple of 128) or [4,E,h,w,56] f16, the level-planar channel-last output of
        droid_backends.corr_pyramid_lookup_nhwc (channel order: corr_channel_map), flow [E,h,w,8] f16 (4 + zero pad),
        ii [E] int64.  Context features, one of
          * inp [E,h,w,128] f16 per edge (the reference's calling convention: any values), or
          * inp = None, inp_frames [K',h,w,128] f16 + inp_index [E] int64: edge e uses row inp_index[e] (what the reference's
            callers pass in effect: video.inps[ii]); the gate convolutions then run over 320 instead of 448 channels on top
            of one per-frame convolution (`context_term`, passed in as `ctx` [K',h,w,384] f32 if the caller keeps it).
        corr0 [E,h,w,128] f16 (then corr is ignored): the output of the correlation encoder's first layer, as
        CorrBlock.lookup_corr0 produces it inside the lookup kernel.
        Round 6, the global-context reduction across iterations (gru.py:23-24 reads the state the previous iteration wrote):
          glo_next = True: the q gate's launch also reduces the NEW hidden state (csrc/conv.hip ConvParams::glo_red); the sums are left in
            self.last_glo ([E,128] f32, or None where fuses_next_glo() is False);
          glo_red = those sums, handed back by a caller that KNOWS `net` is still the tensor the previous call wrote (FactorGraph keeps the
            pair together): the stand-alone reduction -- a pass over the whole hidden state -- is then not launched.  Same values up to the
            order of the per-tile atomics, which is not fixed in either form.
        want_upmask = False: GraphAgg's upmask head (1x1, 128 -> 576; only DepthVideo.upsample reads it, and the reference's callers
        build their graphs with upsample=False unless asked: droid_frontend.py:18, droid_backend.py:30) is not computed; `upmask` is None.
        -> (net, delta [E,h,w,2] f32, weight [E,h,w,2] f32, eta [K,h,w] f32, upmask [K,h,w,576] f16)."""
        P = self.params
        E, h, w, _ = net.shape
        self.last_glo = None
        if self.canvas and not (w == 64 and h % 4 == 0) and w <= 64 and _mask is None:
            return self._forward_canvas(net, inp, corr, flow, ii, inp_frames, inp_index, corr0, _glo=_glo)      # (always with the upmask head)
        if self.canvas and w > 64 and h <= 64 and _mask is None:
            return self._forward_transposed(net, inp, corr, flow, ii, inp_frames, inp_index, corr0)
        if self.canvas and w > 64 and h > 64 and _mask is None:
            return self._forward_strips(net, inp, corr, flow, ii, inp_frames, inp_index, corr0)
        # canvas mode (_mask = (h_img, w_img)): the tensors are canvases; every activation that feeds a 3x3 layer gets the pixels
        # outside the image zeroed again, which is the zero border the reference's padded convolutions see there
        mk = (lambda t: t) if _mask is None else (lambda t: (droid_backends.canvas_mask_(t, _mask[0], _mask[1]), t)[1])
        npix = float(h * w) if _mask is None else float(_mask[0] * _mask[1])
        if _glo is not None:                   # (strip mode: the global context is a mean over the WHOLE image, reduced by the caller)
            glo_red, npix = _glo[0], float(_glo[1])
        main, side = torch.cuda.current_stream(net.device), (None, None)
        if self.streams and net.is_cuda and E * h * w >= (1 << 18):          # (small problems are launch-bound: nothing to overlap)
            if self._side is None:
                self._side = (torch.cuda.Stream(device=net.device), torch.cuda.Stream(device=net.device))
            side = self._side
        c0 = corr0 if corr0 is not None else self.corr0_layer(corr)
        if inp is None and not (w == 64 and h % 4 == 0):
            inp = inp_frames[inp_index]        # image shape outside the production kernel: the reference's data flow
        with _Fork(side[0], main) as fb:       # flow encoder
            f = mk(P["flow2"]([mk(P["flow0"]([flow], EPI_RELU))], EPI_RELU))
        with _Fork(side[1], main) as gb:
            # global context: mean over pixels of sigmoid(w(net)) * net, then three 128x128 GEMVs per edge
            if glo_red is not None and (_mask is None or _glo is not None):
                red = glo_red
            else:
                red = torch.zeros(E, 128, dtype=torch.float32, device=net.device)
                P["gru_w"]([net], EPI_GLO, aux0=net, red=red)
            gzr = droid_backends.glo_gemv(red, P["glo_zr_t"][0], P["glo_zr_t"][1], 1.0 / npix)     # [E,256] z | r
            gq = droid_backends.glo_gemv(red, P["glo_q_t"][0], P["glo_q_t"][1], 1.0 / npix)        # [E,128]
            if inp is None and ctx is None:
                ctx = self.context_term(inp_frames)
        c = mk(P["corr2"]([mk(c0)], EPI_RELU))
        fb.join(f)
        gb.join(gzr, gq, ctx)
        glo = None
        if glo_next and _mask is None and self.fuses_next_glo(h, w):
            self.last_glo = torch.zeros(E, 128, dtype=torch.float32, device=net.device)
            glo = (P["gru_w"].w, P["gru_w"].b, self.last_glo)
        if inp is not None:
            zr = P["zr"]([net, inp, c, f], EPI_GRU_ZR, gterm=gzr, aux0=net)             # [E,h,w,256] = z | r*net
            P["q"]([zr[..., 128:], inp, c, f], EPI_GRU_Q, out=net, gterm=gq, aux0=net, aux1=zr, glo=glo)
        else:
            zr = P["zr_e"]([net, c, f], EPI_GRU_ZR, gterm=gzr, aux0=net, cinit=ctx, cinit_idx=inp_index, cinit_off=0)
            P["q_e"]([zr[..., 128:], c, f], EPI_GRU_Q, out=net, gterm=gq, aux0=net, aux1=zr,
                     cinit=ctx, cinit_idx=inp_index, cinit_off=256, glo=glo)
        mk(net)
        with _Fork(side[0], main) as ab:
            # GraphAgg: conv -> mean over the edges of each source frame -> conv -> eta / upmask
            x = P["agg1"]([net], EPI_RELU)
            order, seg_off = self.segments(ii)
            xm = mk(droid_backends.segment_mean(x, order, seg_off))
            K = xm.shape[0]
            if (_mask is None and w == 64 and h % 4 == 0 and droid_backends.get_option("conv_halo") and droid_backends.get_option("conv_halo2")
                    and droid_backends.get_option("conv_epi_staged") and droid_backends.get_option("eta_fused")):
                # the eta head never runs as a convolution of its own: agg.conv2's workgroups multiply their relu'd tile with its nine
                # tap vectors (the heads' fused second layer), and the gather adds the taps and applies 0.01 * softplus
                x2 = torch.empty(K, h, w, 128, dtype=torch.float16, device=net.device) if want_upmask else None     # (only the upmask head reads it)
                part = torch.empty(1, K * h // 4, 6, 64, 4, dtype=torch.float32, device=net.device)
                P["agg2"]([xm], EPI_HEADS0, out=x2, aux1=P["eta_fused"][0], red=part)
                eta = droid_backends.heads_gather(part, P["eta_fused"][1], h, w, 1)[..., None]
            else:
                x2 = mk(P["agg2"]([xm], EPI_RELU))
                eta = torch.empty(K, h, w, 1, dtype=torch.float32, device=net.device)
                P["eta"]([x2], EPI_SOFTPLUS_001, out=eta)
            upmask = P["upmask"]([x2], EPI_LINEAR) if want_upmask else None
        if _mask is None and w == 64 and h % 4 == 0 and droid_backends.get_option("conv_halo") and droid_backends.get_option("conv_halo2"):
            # heads: the 256-channel activations never leave the first layer's kernel (see csrc/conv.hip EPI_HEADS0)
            w2p, b4 = P["heads2_fused"]
            part = torch.empty(2, E * h // 4, 6, 64, 4, dtype=torch.float32, device=net.device)     # per tile of four rows: output rows -1 .. 4
            P["heads0"]([net], EPI_HEADS0, aux1=w2p, red=part)
            dw = droid_backends.heads_gather(part, b4, h, w)
        else:
            hd = mk(P["heads0"]([net], EPI_RELU))       # (canvas mode: the second layer must see zeros outside the image, so no fusion)
            dw = torch.empty(E, h, w, 4, dtype=torch.float32, device=net.device)
            P["heads2"]([hd], EPI_HEADS, out=dw)
        self.last_dw = dw                        # (delta_x, delta_y, w_x, w_y) as one tensor for droid_backends.ba_inputs
        ab.join(eta, upmask)
        return net, dw[..., :2], dw[..., 2:], eta[..., 0], upmask

    def _cached(self, slot, src, fn):
        """fn(src), kept for as long as `src` is the same tensor with the same version counter (FactorGraph._context hands the same
        frame-level context tensor to every update iteration between two keyframe changes).  `src` is kept alive with the entry, so
        its address cannot be handed to another tensor while the entry exists; inference-mode tensors have no version: recomputed."""
        if src is None:
            return None
        key = tensor_cache_key(src)
        hit = self._derived.get(slot)
        if key is not None and hit is not None and hit[0] == key:
            return hit[1]
        val = fn(src)
        self._derived[slot] = (key, val, src)
        return val

    def _forward_canvas(self, net, inp, corr, flow, ii, inp_frames, inp_index, corr0, _glo=None):
        """forward_nhwc for an image that is not 64 pixels wide / a multiple of four rows high (TUM's 30x40, 16x32, ...): the
        tensors are embedded into zero-padded canvases [.., ceil4(h), 64, C], the production kernels run on the canvases with
        the padding re-zeroed between the layers (forward_nhwc, _mask), and the results are cropped.  Same arithmetic per
        image pixel as on a 64-wide image; the hidden state is still updated in place."""
        E, h, w, _ = net.shape
        Hc, Wc = (h + 3) // 4 * 4, 64
        pad = lambda t: None if t is None else F.pad(t, (0, 0, 0, Wc - w, 0, Hc - h)).contiguous()          # [N,h,w,C] -> [N,Hc,Wc,C]
        crop = lambda t: t[:, :h, :w].contiguous()
        net_c = pad(net)
        if corr0 is not None:
            corr_c, corr0_c = None, pad(corr0)
        elif corr.dim() == 4 and corr.shape[1] == COR_PLANES:                         # [E,196,h,w]
            corr_c, corr0_c = F.pad(corr, (0, Wc - w, 0, Hc - h)).contiguous(), None
        else:                                                                         # [4,E,h,w,56]
            corr_c, corr0_c = F.pad(corr, (0, 0, 0, Wc - w, 0, Hc - h)).contiguous(), None
        # the frame-level context features and the gates' context term depend on the keyframes only: the padded copy and its
        # convolution are kept across the update iterations (round 4 recomputed both on every call for every non-64-wide image)
        inpf_c = self._cached("canvas_inp_%dx%d" % (h, w), inp_frames, pad)
        ctx = self._cached("canvas_ctx_%dx%d" % (h, w), inpf_c, self.context_term) if (inp is None and inpf_c is not None) else None
        n, delta, weight, eta, upmask = self.forward_nhwc(net_c, pad(inp), corr_c, pad(flow), ii, inp_frames=inpf_c,
                                                          inp_index=inp_index, ctx=ctx, corr0=corr0_c, _mask=(h, w), _glo=_glo)
        net.copy_(n[:, :h, :w])
        self.last_dw = crop(self.last_dw)
        return net, self.last_dw[..., :2], self.last_dw[..., 2:], eta[:, :h, :w].contiguous(), crop(upmask)

    # ---- images with more than 64 columns AND more than 64 rows (72x96 = a 576x768 video, ...) ---------------------------------
    STRIP_OVERLAP = 9       # columns a strip's results are contaminated from an INNER strip border: the longest 3x3 / 7x7 chain of one
                            # call is flow_encoder (3 + 1) -> z|r (1) -> q (1) -> agg.conv1 (1) -> agg.conv2 (1) -> eta (1) = 9
                            # (heads: 6 + 1 + 1 = 8; the new hidden state: 6)

    @classmethod
    def strip_plan(cls, w):
        """64-column strips that cover w > 64 columns: [(start, first valid column, one past the last valid column)] -- a strip's results
        are used outside STRIP_OVERLAP columns from its borders that are not image borders"""
        V, plan, lo = cls.STRIP_OVERLAP, [], 0
        while True:
            start = 0 if lo == 0 else lo - V
            if start + 64 >= w:
                start = w - 64
                plan.append((start, lo, w))
                return plan
            plan.append((start, lo, start + 64 - V))
            lo = start + 64 - V

    def _forward_strips(self, net, inp, corr, flow, ii, inp_frames, inp_index, corr0):
        """forward_nhwc for an image wider AND higher than 64 (round 6; rounds 1-5 ran these sizes through the generic convolution loop).
        The production kernels want 64-column images, and every layer of the operator is local except the ConvGRU's global context.  So:
        the global-context sums are reduced over the WHOLE image first (stand-alone kernel, any size); the image is cut into 64-column
        strips that overlap by 2 x STRIP_OVERLAP columns; the strips of all edges run through forward_nhwc as ONE batch of 64-column
        images (each strip's zero padding at an inner border is wrong, and what it contaminates -- at most STRIP_OVERLAP columns, the
        receptive field of the longest convolution chain -- is thrown away); the valid columns are stitched.  GraphAgg's mean over the
        edges of a source frame is per pixel, so frame k of strip s is segment s * K + k of the batch.  Same arithmetic per valid pixel
        as on a 64-column image; 64 / (64 - 2 x 9) more work than the image has."""
        P = self.params
        E, h, w, _ = net.shape
        plan = self.strip_plan(w)
        S = len(plan)
        red = torch.zeros(E, 128, dtype=torch.float32, device=net.device)
        P["gru_w"]([net], EPI_GLO, aux0=net, red=red)
        c0 = corr0 if corr0 is not None else self.corr0_layer(corr)
        cut = lambda t: None if t is None else torch.cat([t[:, :, a:a + 64] for a, _, _ in plan], 0).contiguous()      # [S*N,h,64,C], strip-major
        uniq, ix = torch.unique(ii, return_inverse=True)
        K = uniq.numel()
        ii_b = torch.cat([ix + s_ * K for s_ in range(S)]).contiguous()
        kw = {}
        if inp is None:
            Kp = inp_frames.shape[0]
            kw = dict(inp_frames=self._cached("strips_inp_%dx%d" % (h, w), inp_frames, cut),
                      inp_index=torch.cat([inp_index + s_ * Kp for s_ in range(S)]).contiguous())
            if h % 4 == 0:                     # (other heights: the strips run on canvases, which keep their own padded copy and term)
                kw["ctx"] = self._cached("strips_ctx_%dx%d" % (h, w), kw["inp_frames"], self.context_term)
        net_b = cut(net)
        n_b, _, _, eta_b, up_b = self.forward_nhwc(net_b, cut(inp), None, cut(flow), ii_b, corr0=cut(c0), _glo=(red.repeat(S, 1), h * w), **kw)
        dw_b = self.last_dw
        dw = torch.empty(E, h, w, 4, dtype=dw_b.dtype, device=net.device)
        eta = torch.empty(K, h, w, dtype=eta_b.dtype, device=net.device)
        upmask = torch.empty(K, h, w, up_b.shape[-1], dtype=up_b.dtype, device=net.device)
        for s_, (a, lo, hi) in enumerate(plan):
            net[:, :, lo:hi] = n_b[s_ * E:(s_ + 1) * E, :, lo - a:hi - a]
            dw[:, :, lo:hi] = dw_b[s_ * E:(s_ + 1) * E, :, lo - a:hi - a]
            eta[:, :, lo:hi] = eta_b[s_ * K:(s_ + 1) * K, :, lo - a:hi - a]
            upmask[:, :, lo:hi] = up_b[s_ * K:(s_ + 1) * K, :, lo - a:hi - a]
        self.last_dw, self.last_glo = dw, None
        return net, dw[..., :2], dw[..., 2:], eta, upmask

    # ---- images wider than 64 columns but at most 64 rows high (41x73 from a 16:9 video, 60x80, ...) -------------------------
    def transposed_twin(self):
        """The same operator for the TRANSPOSED image: every k x k kernel transposed (a convolution commutes with swapping the two
        image axes when its taps are swapped too; 1x1 layers and all channel meanings are untouched) and the 7x7 lookup window of
        corr_encoder.0's input channels transposed (a pyramid built from transposed features delivers window sample (xoff, yoff)
        where the original delivers (yoff, xoff)).  An h x w image with w > 64 >= h then runs as the w x h image on the 64-column
        canvases of the production kernels."""
        if self._twin is None:
            self._twin = UpdateModule(self.device, share_inp_by_source_frame=self.share_inp_by_source_frame,
                                      canvas=True).load_state_dict(transposed_state_dict(self._sd))
        return self._twin

    @staticmethod
    def transpose_corr(corr):
        """correlation features of the image -> those of the transposed image (image axes and window axes swapped):
        [E,196,h,w] (channel = level*49 + xoff*7 + yoff) or the level-planar [4,E,h,w,56] (channel = yoff*7 + xoff, 49..55 pad)"""
        if corr.dim() == 4:
            E, _, h, w = corr.shape
            return corr.reshape(E, 4, 7, 7, h, w).permute(0, 1, 3, 2, 5, 4).reshape(E, COR_PLANES, w, h).contiguous()
        k = torch.arange(56, device=corr.device)
        perm = torch.where(k < 49, (k % 7) * 7 + k // 7, k)
        return corr.transpose(2, 3)[..., perm].contiguous()

    def _forward_transposed(self, net, inp, corr, flow, ii, inp_frames, inp_index, corr0):
        """forward_nhwc for an image with more than 64 columns and at most 64 rows: run the transposed twin on the transposed tensors
        (it embeds them into [.., ceil4(w), 64, C] canvases, _forward_canvas) and transpose the results back."""
        twin = self.transposed_twin()
        t = lambda x: None if x is None else x.transpose(1, 2).contiguous()                 # [N,h,w,C] <-> [N,w,h,C]
        corr_t = None if (corr is None or corr0 is not None) else self.transpose_corr(corr)
        inpf_t = self._cached("transposed_inp", inp_frames, t)       # (a stable tensor: the twin's canvas copy / context term then hit too)
        n, _, _, eta, upmask = twin.forward_nhwc(t(net), t(inp), corr_t, t(flow), ii, inp_frames=inpf_t, inp_index=inp_index,
                                                 corr0=t(corr0))
        net.copy_(n.transpose(1, 2))
        self.last_dw = t(twin.last_dw)
        return net, self.last_dw[..., :2], self.last_dw[..., 2:], eta.transpose(1, 2).contiguous(), t(upmask)

    def corr_to_nhwc(self, corr):
        """[E,196,h,w] correlation features in the reference's channel order (level*49 + xoff*7 + yoff) -> the level-planar
        channel-last [4,E,h,w,56] tensor that forward_nhwc consumes (what corr_pyramid_lookup_nhwc writes directly)"""
        E, _, ht, wd = corr.shape
        cpad = torch.cat([corr, torch.zeros_like(corr[:, :1])], 1)              # channel 196 = zeros for the pads
        c = self.to_nhwc(cpad[:, torch.where(self.cmap >= 0, self.cmap, torch.full_like(self.cmap, COR_PLANES))])
        return c.view(E, ht, wd, 4, 56).permute(3, 0, 1, 2, 4).contiguous()

    def forward(self, net, inp, corr, flow=None, ii=None, jj=None):
        """Reference interface (droid_net.py:111-143): net, inp [1,E,128,h,w], corr [1,E,196,h,w],
        flow [1,E,4,h,w] -> net [1,E,128,h,w], delta, weight [1,E,h,w,2], eta [1,K,h,w], upmask [1,K,576,h,w]."""
        batch, num, ch, ht, wd = net.shape
        assert batch == 1
        if flow is None:
            flow = torch.zeros(batch, num, 4, ht, wd, device=net.device)
        if ii is None:
            ii = torch.arange(num, device=net.device)
        n = self.to_nhwc(net[0]); i = self.to_nhwc(inp[0])
        if n.data_ptr() == net.data_ptr():
            # a channel-last fp16 input (what the encoders of this library produce) makes to_nhwc a view: forward_nhwc
            # updates its hidden state in place, the reference interface must not touch the caller's tensor
            n = n.clone()
        c = corr[0].half().contiguous() if self.wants_reference_layout_corr(ht, wd) else self.corr_to_nhwc(corr[0])
        f = self.to_nhwc(flow[0], 8)
        ii = ii.to(net.device)
        if self.share_inp_by_source_frame:
            # edges with equal ii carry the same context features (true for every call site of the reference:
            # factor_graph.py:135,299, motion_filter.py): convolve them once per source frame
            uniq, ix = torch.unique(ii, return_inverse=True)
            first = torch.full((uniq.numel(),), ii.numel(), dtype=torch.long, device=ii.device)
            first.scatter_reduce_(0, ix, torch.arange(ii.numel(), device=ii.device), reduce="amin")
            n, delta, weight, eta, upmask = self.forward_nhwc(n, None, c, f, ii, inp_frames=i[first].contiguous(), inp_index=ix.contiguous())
        else:
            n, delta, weight, eta, upmask = self.forward_nhwc(n, i, c, f, ii)
        net_out = n.permute(0, 3, 1, 2)[None].to(net.dtype)
        return net_out, delta[None], weight[None], eta[None], upmask.permute(0, 3, 1, 2)[None]

    __call__ = forward
