from conformer_amd.model.utils.masking import generate_padding_mask  # noqa: F401
