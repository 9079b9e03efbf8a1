"""conformer_amd: MI355X-native (gfx950) Conformer encoder hot path behind the reference's nn.Module surface."""
__version__ = "0.1.0"


def invalidate_weight_caches() -> None:
    """Drop every derived weight tensor (see conformer_amd.model.utils._guard.invalidate_weight_caches): needed only after
    writes that bypass PyTorch's version counter."""
    from conformer_amd.model.utils._guard import invalidate_weight_caches as _inv
    _inv()
