"""Feature / context encoders on MI355X: host-side mirror of the reference's BasicEncoder (droid_slam/modules/extractor.py:
120-198; ResidualBlock :6-56) as DROID-SLAM instantiates it (droid_net.py:149-150: fnet = BasicEncoder(128, 'instance'),
cnet = BasicEncoder(256, 'none')) and of DroidNet.extract_features' image normalisation (droid_net.py:154-171).

They produce what the hot path consumes -- fmaps / nets / inps [buf,*,128,h/8,w/8] fp16 -- once per incoming frame
(~11 GFLOP each at 384x512).  Every convolution runs in the implicit-GEMM MFMA kernel of this library
(droid_backends.conv2d_nhwc, channel-last fp16, fp32 accumulation); the stride-2 layers (stem, the first convolution and the 1x1
shortcut of layer2 / layer3) through its stride-2 form (droid_backends.conv2d_s2_nhwc: output pixel (y, x) reads input (2y + dy - pad,
2x + dx - pad) -- the even positions of the stride-1 result, which is what rounds 3-5 computed in full and sliced); instance normalisation,
residual additions and activations are droid_backends.norm_act.  Accepts the reference's parameter names
(`conv1.weight`, `layer2.0.downsample.0.weight`, ...), so a droid.pth checkpoint loads as is.
"""
import torch

import droid_backends as db
from .update import _Conv, pack_conv, pack_conv_halo, EPI_LINEAR, EPI_RELU

DIM = 32
MEAN = (0.485, 0.456, 0.406)
STDV = (0.229, 0.224, 0.225)


def reference_param_shapes(output_dim):
    """state_dict names -> (Cout, Cin, k) of BasicEncoder(output_dim) (extractor.py:140-151, 181-187)"""
    shapes = {"conv1": (DIM, 3, 7), "conv2": (output_dim, 4 * DIM, 1)}
    cin = DIM
    for li, (dim, stride) in enumerate(((DIM, 1), (2 * DIM, 2), (4 * DIM, 2)), start=1):
        for bi in range(2):
            p = "layer%d.%d." % (li, bi)
            shapes[p + "conv1"] = (dim, cin if bi == 0 else dim, 3)
            shapes[p + "conv2"] = (dim, dim, 3)
            if bi == 0 and stride != 1:
                shapes[p + "downsample.0"] = (dim, cin, 1)
        cin = dim
    return shapes


def empty_state_dict(output_dim, dtype=torch.float32):
    sd = {}
    for k, (co, ci, ks) in reference_param_shapes(output_dim).items():
        sd[k + ".weight"] = torch.zeros(co, ci, ks, ks, dtype=dtype)
        sd[k + ".bias"] = torch.zeros(co, dtype=dtype)
    return sd


class BasicEncoder:
    def __init__(self, output_dim=128, norm_fn="instance", device="cuda"):
        assert norm_fn in ("instance", "none"), "DROID-SLAM uses 'instance' (fnet) and 'none' (cnet) (droid_net.py:149-150)"
        self.output_dim, self.norm, self.device = output_dim, norm_fn == "instance", torch.device(device)
        self.params = None

    def load_state_dict(self, sd, prefix=""):
        P = {}
        for name, (co, ci, k) in reference_param_shapes(self.output_dim).items():
            w = sd[prefix + name + ".weight"].to(self.device); b = sd[prefix + name + ".bias"].to(self.device)
            cin_pad = 8 if ci == 3 else None
            P[name] = _Conv(*pack_conv(w, b, cin_pad), k, co, pack_conv_halo(w) if cin_pad is None else None)
        self.params = P
        return self

    # conv -> [instance norm] -> relu ; the convolution's own epilogue does the relu when there is no normalisation
    def _cnr(self, name, x, stride=1, relu=True):
        fuse = relu and not self.norm
        p = self.params[name]
        if stride == 2:
            # native stride 2 (droid_backends.conv2d_s2_nhwc): the even positions of the stride-1 result at a quarter of its work, no
            # strided copy of the output (round 5 computed the stride-1 result -- 4.9 GFLOP for the 0.46 GFLOP stem -- and sliced it)
            y = db.conv2d_s2_nhwc(x.contiguous(), p.w, p.b, p.k, p.k, p.cout, EPI_RELU if fuse else EPI_LINEAR)
        else:
            y = p([x], EPI_RELU if fuse else EPI_LINEAR)
        if self.norm or (relu and not fuse):
            y = db.norm_act(y, None, self.norm, relu)
        return y

    def _block(self, prefix, x, stride):
        y = self._cnr(prefix + "conv1", x, stride)
        y = self._cnr(prefix + "conv2", y, 1)
        if stride != 1:
            x = self._cnr(prefix + "downsample.0", x, 2, relu=False)             # 1x1 / 2: reads the even pixels itself
        return db.norm_act(x, y, False, True)                                   # relu(x + y)

    def forward(self, x):
        """x [B,N,3,H,W] normalised image (float) -> [B,N,output_dim,H/8,W/8] fp16"""
        b, n, c, h, w = x.shape
        assert c == 3 and h % 8 == 0 and w % 8 == 0
        t = torch.zeros(b * n, h, w, 8, dtype=torch.float16, device=self.device)
        t[..., :3] = x.reshape(b * n, 3, h, w).permute(0, 2, 3, 1)
        t = self._cnr("conv1", t, 2)
        for li, stride in ((1, 1), (2, 2), (3, 2)):
            t = self._block("layer%d.0." % li, t, stride)
            t = self._block("layer%d.1." % li, t, 1)
        t = self.params["conv2"]([t], EPI_LINEAR)
        return t.permute(0, 3, 1, 2).reshape(b, n, self.output_dim, h // 8, w // 8)

    __call__ = forward


def normalize_images(images):
    """uint8 / float BGR images [B,N,3,H,W] -> normalised RGB (droid_net.py:157-161, motion_filter.py:66-67)"""
    x = images[:, :, [2, 1, 0]].float() / 255.0
    mean = torch.as_tensor(MEAN, device=x.device)[:, None, None]
    std = torch.as_tensor(STDV, device=x.device)[:, None, None]
    return (x - mean) / std


class FeatureNets:
    """fnet + cnet + the split of the context output (droid_net.py:154-171): images -> (fmaps, net, inp)"""

    def __init__(self, device="cuda"):
        self.fnet = BasicEncoder(128, "instance", device)
        self.cnet = BasicEncoder(256, "none", device)

    def load_state_dict(self, sd, prefix=""):
        self.fnet.load_state_dict(sd, prefix + "fnet.")
        self.cnet.load_state_dict(sd, prefix + "cnet.")
        return self

    def extract_features(self, images):
        x = normalize_images(images)
        fmaps = self.fnet(x)
        net, inp = self.cnet(x).split([128, 128], dim=2)
        return fmaps, torch.tanh(net.float()).half(), torch.relu(inp)
