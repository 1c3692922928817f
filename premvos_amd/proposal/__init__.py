from .model import ProposalNet, cell_anchors  # noqa: F401
from .driver import (OfflinePredictor, ProposalStage, convert_results_to_json, custom_resize_shape,  # noqa: F401
                     detect_one_image, forward)
