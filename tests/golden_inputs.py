"""Seeded inputs shared by tests/golden/make_golden.py (which runs the REFERENCE on them) and the tests (which run the
oracle / the HIP path on the same values): keeps the committed fixtures to the reference's OUTPUTS."""
import numpy as np
import torch

UPDATE_AUTOCAST = dict(E=3, ht=8, wd=64, seed=41, ii=[0, 0, 2], jj=[1, 2, 0], weight_seed=1234)


def update_autocast_inputs():
    """fp16-representable inputs of one UpdateModule.forward as FactorGraph.update feeds it under autocast
    (factor_graph.py:214-231): net, inp, corr in fp16, motion features fp32 clamped to +-64.  W = 64 and H % 4 == 0, so
    the HIP path takes its production convolution kernels."""
    c = UPDATE_AUTOCAST
    rng = np.random.default_rng(c["seed"])
    E, ht, wd = c["E"], c["ht"], c["wd"]
    net = np.tanh(rng.standard_normal((E, 128, ht, wd))).astype(np.float16)
    inp = np.maximum(rng.standard_normal((E, 128, ht, wd)), 0).astype(np.float16)
    corr = (2.0 * rng.standard_normal((E, 196, ht, wd))).astype(np.float16)
    flow = np.clip(4.0 * rng.standard_normal((E, 4, ht, wd)), -64, 64).astype(np.float16).astype(np.float32)
    t = lambda a: torch.as_tensor(a)
    return t(net), t(inp), t(corr), t(flow), torch.as_tensor(c["ii"]), torch.as_tensor(c["jj"])
