from busca_amd.option import *  # noqa: F401,F403
from busca_amd.option import load_args_from_config, merge_args  # noqa: F401
