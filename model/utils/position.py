from conformer_amd.model.utils.position import RelativePositionalEncoding  # noqa: F401
