"""CPU checks of the algebra behind the transposed path for images with more than 64 columns (droid_amd/corr.py CorrBlock.transposed,
droid_amd/update.py UpdateModule.transposed_twin), on the oracle (the CPU restatement of the reference, oracle/):

  * the update operator with `transposed_state_dict` on transposed inputs == the operator on the inputs, transposed
    (droid_net.py:78-143: every layer is a convolution, an elementwise function, a mean over pixels or a mean over edges);
  * the 4-level lookup of transposed features at transposed coordinates == the lookup, with image AND window axes swapped
    (modules/corr.py:23-50, correlation_kernels.cu:20-71), floor pooling included (odd sizes).
"""
import numpy as np
import torch

from oracle import corr as ocorr, update as oupd
from droid_amd.update import UpdateModule, transposed_state_dict
from droid_amd.weights import deterministic_state_dict


class _SD:
    def state_dict(self):
        return oupd.empty_state_dict()


def test_update_operator_commutes_with_transposition():
    torch.manual_seed(0)
    sd = deterministic_state_dict(_SD(), seed=11)
    E, h, w = 3, 5, 9
    net = torch.tanh(torch.randn(E, 128, h, w)); inp = torch.relu(torch.randn(E, 128, h, w))
    corr = torch.randn(E, 196, h, w); flow = torch.randn(E, 4, h, w)
    ii = torch.tensor([0, 0, 1])
    ref = oupd.update_forward(sd, net, inp, corr, flow, ii)
    t = lambda x: x.transpose(-1, -2).contiguous()                      # [.., h, w] -> [.., w, h]
    corr_t = UpdateModule.transpose_corr(corr)
    assert corr_t.shape == (E, 196, w, h)
    # channel l*49 + a*7 + b of the transposed features at (x, y) is channel l*49 + b*7 + a of the features at (y, x)
    assert torch.equal(corr_t[1, 2 * 49 + 3 * 7 + 5, 4, 2], corr[1, 2 * 49 + 5 * 7 + 3, 2, 4])
    got = oupd.update_forward(transposed_state_dict(sd), t(net), t(inp), corr_t, t(flow), ii)
    n, delta, weight, eta, upmask = got
    back = (t(n), delta.transpose(1, 2), weight.transpose(1, 2), t(eta), t(upmask))
    for a, b in zip(back, ref):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
    # the level-planar channel-last form [4,E,h,w,56] (channel = yoff*7 + xoff, 49..55 padding)
    nh = torch.randn(4, E, h, w, 56)
    nt = UpdateModule.transpose_corr(nh)
    assert nt.shape == (4, E, w, h, 56)
    assert torch.equal(nt[2, 1, 4, 2, 3 * 7 + 5], nh[2, 1, 2, 4, 5 * 7 + 3]) and torch.equal(nt[..., 49:], nh.transpose(2, 3)[..., 49:])


def test_lookup_of_transposed_features_is_the_transposed_lookup():
    rng = np.random.default_rng(3)
    E, h, w = 2, 9, 14                                                   # odd / non-multiple-of-8 sizes: floor pooling on every level
    f1 = rng.standard_normal((E, 16, h, w)).astype(np.float32); f2 = rng.standard_normal((E, 16, h, w)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    coords = np.stack([xx, yy], -1)[None] + rng.uniform(-3, 3, (E, 1, 1, 2)).astype(np.float32) + 0.1 * yy[None, :, :, None]
    coords = coords.astype(np.float32)                                   # [E,h,w,2] (x, y)
    ref = ocorr.corr_block_lookup(ocorr.corr_pyramid(f1, f2, 4), coords, 3)                       # [E,196,h,w]
    f1t, f2t = np.ascontiguousarray(f1.transpose(0, 1, 3, 2)), np.ascontiguousarray(f2.transpose(0, 1, 3, 2))
    coords_t = np.ascontiguousarray(coords.transpose(0, 2, 1, 3)[..., ::-1])                     # pixel (x, y) looks at (y', x')
    out_t = ocorr.corr_block_lookup(ocorr.corr_pyramid(f1t, f2t, 4), coords_t, 3)                # [E,196,w,h]
    back = out_t.reshape(E, 4, 7, 7, w, h).transpose(0, 1, 3, 2, 5, 4).reshape(E, 196, h, w)     # CorrBlock.__call__, transposed
    assert np.abs(back - ref).max() <= 1e-5 * np.abs(ref).max()


def test_lookup_is_the_sum_of_the_lookups_on_target_strips():
    """the algebra behind CorrBlock.strips (images with more than 64 columns and rows), on the oracle: the correlation volume is
    linear in the target features and a lookup reads zero outside its volume (correlation_kernels.cu:48 within_bounds), so the
    4-level lookup of a source strip equals the SUM over disjoint target strips of the lookups in the strips' own pyramids at
    coordinates shifted by the strip origin -- windows that straddle a strip border included, and floor pooling included as long as
    the strips start at multiples of 2^3 (here: strips of 8 and 16 columns on a 12 x 27 image; the product uses 64)."""
    rng = np.random.default_rng(5)
    E, C, h, w = 2, 16, 12, 27
    f1 = rng.standard_normal((E, C, h, w)).astype(np.float32); f2 = rng.standard_normal((E, C, h, w)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    coords = (np.stack([xx, yy], -1)[None] + rng.uniform(-4, 4, (E, 1, 1, 2)).astype(np.float32) + 0.13 * yy[None, :, :, None]).astype(np.float32)
    ref = ocorr.corr_block_lookup(ocorr.corr_pyramid(f1, f2, 4), coords, 3)                       # [E,196,h,w]

    def pyramid(a, b):                                             # a [E,C,h,wa], b [E,C,h,wb] -> 4 levels of [E,h,wa,h2,wb2]
        va = a.reshape(E, C, -1).astype(np.float64) / 4.0; vb = b.reshape(E, C, -1).astype(np.float64) / 4.0
        vol = np.einsum("ecp,ecq->epq", va, vb).reshape(E, a.shape[2], a.shape[3], b.shape[2], b.shape[3])
        out = []
        for _ in range(4):
            out.append(vol); vol = ocorr.avg_pool2(vol)
        return out

    for bounds in ([(0, 8), (8, 8), (16, 11)], [(0, 16), (16, 11)]):                               # target (and source) strips (x0, width)
        cols = []
        for (xs, ws) in bounds:
            acc = 0.0
            for (xt, wt) in bounds:
                pyr = pyramid(f1[..., xs:xs + ws], f2[..., xt:xt + wt])
                c = coords[:, :, xs:xs + ws].copy(); c[..., 0] -= xt
                cc = np.ascontiguousarray(np.moveaxis(c, -1, 1)).astype(np.float32)
                acc = acc + np.concatenate([ocorr.corr_index_forward(v, (cc / np.float32(2 ** l)).astype(np.float32), 3).reshape(E, -1, h, ws)
                                            for l, v in enumerate(pyr)], axis=1)
            cols.append(acc)
        got = np.concatenate(cols, -1)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), bounds
