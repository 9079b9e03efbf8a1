from conformer_amd.model.utils.activation import GLU, Swish  # noqa: F401
