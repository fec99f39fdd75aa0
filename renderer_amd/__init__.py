"""renderer_amd — MI355X-native instance pipeline (transform -> world AABB -> frustum cull ->
indirect-draw-command compaction) of farnoy/renderer, behind a C ABI.

The compute path is the HIP library built from renderer_amd/csrc (see include/
mi_instance_pipeline.h). This package is the host-side harness over that ABI: ctypes
binding, synthetic scenes, and the torch.distributed plumbing for sharded scenes. There is
no CPU fallback: without the built library, or without a gfx950 device, it raises.
"""
from ._lib import MipError, load_library, library_path  # noqa: F401
from .pipeline import InstancePipeline, MESH_DTYPE, DRAW_CMD_DTYPE  # noqa: F401
from . import scene  # noqa: F401
