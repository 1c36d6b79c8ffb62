"""rlvi_amd: MI355X-native (gfx950) RLVI E-step / M-step hot path behind the reference's
`--method=rlvi` plug-in interface.  See DESIGN.md and include/rlvi_hip.h."""
import os as _os

__version__ = "0.2.0"

# Several processes on one node (one per GPU) exchange IPC handles of device memory: RCCL for its own
# buffers, rlvi_amd.dist.setup_peers for the inboxes of the sharded E-step.  The driver stack of this
# platform only supports the dmabuf IPC mode, selected by HSA_ENABLE_IPC_MODE_LEGACY=0; without it
# hipIpcGetMemHandle fails with "invalid argument".  The ROCm runtime reads the variable when it starts
# (the first GPU call of the process), so it is defaulted here, at import time, and never overridden.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
