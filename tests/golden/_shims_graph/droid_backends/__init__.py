"""Oracle-backed stand-in for the reference's CUDA extension -- GOLDEN GENERATION ONLY (tests/golden/make_graph_golden.py).

Lets the reference's unmodified factor_graph.py / depth_video.py / modules/corr.py run on CPU: every entry point forwards
to the numpy oracle (which is itself pinned to the reference's CUDA kernels by tests/golden/ref_cuda.npz).  What the
resulting vectors pin is the reference's ORCHESTRATION: index bookkeeping, tensor views, ordering of updates, edge
selection.  Not part of the product."""
import numpy as np
import torch

from oracle import ba as _oba, corr as _ocorr, geom as _ogeom


def _np(t):
    return t.detach().cpu().numpy()


# Switches of the LARGE goldens (tests/golden/make_graph_scale_golden.py c5), off by default:
ALPHA_MAP = None        # [buf,ht,wd] per-pixel weight of the sensor-depth prior (BASELINE configs[4]; the reference's ba has the constant 0.05)
FAST_ALTCORR = False    # oracle.corr.altcorr_forward_fast instead of the per-tap gathers (pinned to them by tests/test_oracle_golden.py)
BA_THREADS = 1


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only):
    p = poses.numpy(); d = disps.numpy()                      # views: updated in place like the CUDA extension does
    dx, dz = _oba.ba(p, d, _np(intrinsics), _np(disps_sens), _np(targets), _np(weights), _np(eta), _np(ii), _np(jj),
                     int(t0), int(t1), int(iterations), float(lm), float(ep), bool(motion_only), dtype=np.float64,
                     alpha_map=ALPHA_MAP, threads=BA_THREADS)
    return [torch.as_tensor(dx, dtype=torch.float32), torch.as_tensor(dz if dz is not None else np.zeros(0), dtype=torch.float32)]


def corr_index_forward(volume, coords, radius):
    outs = []
    for s in range(0, volume.shape[0], 16):           # bounded fp32 copies (a 512-edge level-0 volume is 19 GB in fp32)
        outs.append(torch.as_tensor(_ocorr.corr_index_forward(_np(volume[s:s + 16].float()), _np(coords[s:s + 16].float()), int(radius))).to(volume.dtype))
    return [torch.cat(outs, 0)]


def altcorr_forward(fmap1, fmap2, coords, ii, jj, radius):
    if FAST_ALTCORR:
        out = _ocorr.altcorr_forward_fast(fmap1, fmap2, coords.float(), _np(ii), _np(jj), int(radius))
    else:
        out = _ocorr.altcorr_forward(_np(fmap1.float()), _np(fmap2.float()), _np(coords.float()), _np(ii), _np(jj), int(radius))
    return [torch.as_tensor(out).to(fmap1.dtype)]


def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    d = _ogeom.frame_distance(_np(poses), _np(disps), _np(intrinsics), _np(ii), _np(jj), float(beta))
    return torch.as_tensor(d, dtype=torch.float32)
