from busca_amd.visualization import *  # noqa: F401,F403
from busca_amd.visualization import plot_box, create_batch_image  # noqa: F401
