from conformer_amd.model.utils.ffn import FeedForwardModule  # noqa: F401
