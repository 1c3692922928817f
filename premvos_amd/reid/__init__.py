from .model import ReIDNet, context_boxes  # noqa: F401
from .driver import Config, ReIDEngine, engine_from_config, units_from_config, ReID_net_init, add_ReID, forward_directory  # noqa: F401
