from .model import RefinementNet, module_plan  # noqa: F401
from .driver import Config, DataKeys, Extractions, RefinementEngine, do_refinement, forward_directory, refinement_net_init  # noqa: F401
