from conformer_amd.model.modules.encoder import Encoder  # noqa: F401
