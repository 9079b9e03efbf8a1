from conformer_amd.model.utils.attention import MultiHeadSelfAttentionModule, RelativeMultiHeadAttention  # noqa: F401
