"""Oracle restatement (torch-CPU, functional) of the feature / context encoders.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  basic_encoder ... BasicEncoder.forward + ResidualBlock.forward   droid_slam/modules/extractor.py:6-56, 120-198
Parameters by the reference's state_dict names.  ``autocast=True`` evaluates under torch.autocast(fp16) like the callers do
(motion_filter.py:38-49: @autocast(enabled=True)).  Pinned by tests/golden/encoder_python.npz (the reference's own module).
"""
import torch
import torch.nn.functional as F


def _conv(p, name, x, stride=1):
    w = p[name + ".weight"]
    return F.conv2d(x, w, p[name + ".bias"], stride=stride, padding=w.shape[-1] // 2)


def _norm(x, instance):
    return F.instance_norm(x) if instance else x


def _block(p, prefix, x, stride, instance):
    y = torch.relu(_norm(_conv(p, prefix + "conv1", x, stride), instance))
    y = torch.relu(_norm(_conv(p, prefix + "conv2", y), instance))
    if stride != 1:
        x = _norm(_conv(p, prefix + "downsample.0", x, stride), instance)
    return torch.relu(x + y)


def basic_encoder(p, x, instance, autocast=False):
    """x [M,3,H,W] normalised images -> [M,output_dim,H/8,W/8]"""
    if autocast:
        with torch.autocast("cpu", dtype=torch.float16):
            return basic_encoder(p, x, instance, autocast=False)
    x = torch.relu(_norm(_conv(p, "conv1", x, 2), instance))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = _block(p, "layer%d.0." % li, x, stride, instance)
        x = _block(p, "layer%d.1." % li, x, 1, instance)
    return _conv(p, "conv2", x)
