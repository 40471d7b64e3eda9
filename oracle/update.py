"""Oracle restatement (torch-CPU, functional, fp32/fp64) of the ConvGRU update block.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  update_forward ... UpdateModule.forward   droid_slam/droid_net.py:111-143
  _conv_gru ........ ConvGRU.forward        droid_slam/modules/gru.py:19-33
  _graph_agg ....... GraphAgg.forward       droid_slam/droid_net.py:59-75
  cvx_upsample ..... cvx_upsample           droid_slam/droid_net.py:21-35

Parameters are addressed by the reference's state_dict names (PARAM_SHAPES) so a
``droid.pth``-style checkpoint, the deterministic fill of
``droid_amd.weights`` and the golden vectors all line up.
"""
import numpy as np
import torch
import torch.nn.functional as F

COR_PLANES = 4 * 49

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
    sd = {}
    for k, (co, ci, ks) in PARAM_SHAPES.items():
        sd[k + ".weight"] = torch.zeros(co, ci, ks, ks, dtype=dtype)
        sd[k + ".bias"] = torch.zeros(co, dtype=dtype)
    return sd


def _conv(p, name, x):
    w = p[name + ".weight"]
    return F.conv2d(x, w, p[name + ".bias"], padding=w.shape[-1] // 2)


def _conv_gru(p, net, inp):
    hx = torch.cat([net, inp], 1)
    g = torch.sigmoid(_conv(p, "gru.w", net)) * net
    b, c, h, w = net.shape
    g = g.view(b, c, h * w).mean(-1).view(b, c, 1, 1)
    z = torch.sigmoid(_conv(p, "gru.convz", hx) + _conv(p, "gru.convz_glo", g))
    r = torch.sigmoid(_conv(p, "gru.convr", hx) + _conv(p, "gru.convr_glo", g))
    q = torch.tanh(_conv(p, "gru.convq", torch.cat([r * net, inp], 1)) + _conv(p, "gru.convq_glo", g))
    return (1 - z) * net + z * q


def _graph_agg(p, net, ii):
    _, ix = torch.unique(ii, return_inverse=True)
    x = torch.relu(_conv(p, "agg.conv1", net))
    K = int(ix.max().item()) + 1
    acc = torch.zeros((K,) + x.shape[1:], dtype=x.dtype).index_add_(0, ix, x)
    cnt = torch.zeros(K, dtype=x.dtype).index_add_(0, ix, torch.ones(len(ix), dtype=x.dtype))
    x = acc / cnt.view(-1, 1, 1, 1)
    x = torch.relu(_conv(p, "agg.conv2", x))
    eta = 0.01 * F.softplus(_conv(p, "agg.eta.0", x).float())[:, 0]      # softplus is on autocast's fp32 list on a GPU
    upmask = _conv(p, "agg.upmask.0", x)
    return eta, upmask


def update_forward(p, net, inp, corr, flow, ii, autocast=False):
    """net,inp [E,128,h,w], corr [E,196,h,w], flow [E,4,h,w], ii [E] ->
    (net' [E,128,h,w], delta [E,h,w,2], weight [E,h,w,2], eta [K,h,w], upmask [K,576,h,w]).

    ``autocast=True`` evaluates the same expressions under torch.autocast(fp16) like the reference's caller does
    (factor_graph.py:13-16,214): every convolution output and every elementwise result is stored in fp16 (fp32
    accumulation inside a convolution); net/inp/corr should then be fp16 tensors.  Pinned by
    tests/golden/update_autocast_python.npz (the reference's own module under autocast)."""
    if autocast:
        with torch.autocast("cpu", dtype=torch.float16):
            return update_forward(p, net, inp, corr, flow, ii, autocast=False)
    c = torch.relu(_conv(p, "corr_encoder.2", torch.relu(_conv(p, "corr_encoder.0", corr))))
    f = torch.relu(_conv(p, "flow_encoder.2", torch.relu(_conv(p, "flow_encoder.0", flow))))
    net = _conv_gru(p, net, torch.cat([inp, c, f], 1))
    delta = _conv(p, "delta.2", torch.relu(_conv(p, "delta.0", net)))
    weight = torch.sigmoid(_conv(p, "weight.2", torch.relu(_conv(p, "weight.0", net))))
    eta, upmask = _graph_agg(p, net, ii)
    return net, delta.permute(0, 2, 3, 1).contiguous(), weight.permute(0, 2, 3, 1).contiguous(), eta, upmask


def cvx_upsample(data, mask):
    """data [B,h,w,D], mask [B,576,h,w] -> [B,8h,8w,D]: softmax-weighted 3x3 convex combination."""
    B, h, w, D = data.shape
    m = torch.softmax(mask.reshape(B, 9, 8, 8, h, w), dim=1)
    d = F.pad(data.permute(0, 3, 1, 2), (1, 1, 1, 1))
    nb = torch.stack([d[:, :, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], 2)  # [B,D,9,h,w]
    up = torch.einsum("bkyxhw,bdkhw->bhywxd", m, nb)
    return up.reshape(B, 8 * h, 8 * w, D)
