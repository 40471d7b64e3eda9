"""Dependency-free helper shared by update.py, factor_graph.py and dist_ba.py (no import of the compiled extension)."""


def tensor_cache_key(*tensors):
    """identity + version of tensors a cached derivation depends on, or None when no key can be formed (tensors created
    under torch.inference_mode() carry no version counter: their derivations are then recomputed on every call)"""
    key = []
    for t in tensors:
        if t.is_inference():
            return None
        key.append((t.data_ptr(), t._version, t.numel(), t.device))
    return tuple(key)
