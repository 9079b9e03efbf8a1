"""conformer_amd: MI355X-native (gfx950) Conformer encoder hot path behind the reference's nn.Module surface."""
__version__ = "0.1.0"
