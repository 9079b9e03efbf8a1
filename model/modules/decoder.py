from conformer_amd.model.modules.decoder import Decoder  # noqa: F401
