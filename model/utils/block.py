from conformer_amd.model.utils.block import ConformerBlock  # noqa: F401
