from busca_amd.network import *  # noqa: F401,F403
from busca_amd.network import BUSCA, ReID_Encoder, memory_indices  # noqa: F401
