from .model import ReIDNet, context_boxes  # noqa: F401
from .driver import Config, ReIDEngine, ReID_net_init, add_ReID, forward_directory  # noqa: F401
