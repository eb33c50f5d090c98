"""Invalidation of the packed device copies of weights (bf16 GEMM operands, fused adaLN matrices, decoder fragments).

Every module that keeps such a cache stamps it with the global weights EPOCH; anything that changes parameters bumps the
epoch: `load_state_dict` anywhere in a tree that holds a cache (post-hook on the holder), `Module._apply` (.to / .cuda),
and the in-place writers of this package (`parallel.broadcast_flat`, `synth.fill_module_random_`, `checkpoint.load_checkpoint`).
Callers that write parameters in place themselves call `ln3diff_amd.invalidate_weight_caches()`.
"""
EPOCH = [0]


def bump(*_a, **_k):
    EPOCH[0] += 1


def stamp(cache, owner=None):
    """Mark a freshly built cache with the current epoch.  `owner` (the module the cache belongs to) gets the load hook on EVERY
    submodule it has by now, so that a state dict loaded straight into a child (`dit.blocks[3].load_state_dict(...)`) also drops
    the parent's packed copies - the hook registered in the holder's __init__ only sees loads that go through the holder."""
    cache['epoch'] = EPOCH[0]
    if owner is not None:
        watch_tree(owner)
    return cache


def fresh(cache, device=None, key='device'):
    return cache is not None and cache.get('epoch') == EPOCH[0] and (device is None or cache.get(key) == device)


def watch_tree(module):
    for sub in module.modules():
        if not sub.__dict__.get('_ln3d_watched', False):
            sub.register_load_state_dict_post_hook(bump)
            sub.__dict__['_ln3d_watched'] = True
    return module


def watch(module):
    """Call in the __init__ of a cache-holding module: weights loaded into it (or into a parent) drop every cache."""
    if not module.__dict__.get('_ln3d_watched', False):
        module.register_load_state_dict_post_hook(bump)
        module.__dict__['_ln3d_watched'] = True
    return module
