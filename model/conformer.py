from conformer_amd.model.conformer import *  # noqa: F401,F403
from conformer_amd.model.conformer import Conformer  # noqa: F401
