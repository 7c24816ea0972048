from busca_amd.tracking import *  # noqa: F401,F403
from busca_amd.tracking import center_distance, get_bbox_crop, missing_candidate_bbox, iou_distance  # noqa: F401
