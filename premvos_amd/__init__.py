"""premvos_amd -- MI355X-native (gfx950) hot path of PReMVOS: PWC-Net flow, proposal_net and
refinement_net forward passes on hand-written HIP kernels behind a C-ABI (include/premvos_hip.h)."""
__version__ = "0.1.0"
