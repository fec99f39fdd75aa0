"""CPU oracle of the instance pipeline — TEST INFRASTRUCTURE ONLY (parity unpinned).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package. See mip_oracle.h for the reference citations.
"""
from .oracle import (  # noqa: F401
    ORC_MESH_DTYPE,
    DRAW_CMD_DTYPE,
    build,
    lib,
    run,
    Runner,
    model_matrix,
    world_aabb,
    coarse_culled,
    pick_lod,
    project_camera,
    compact_draw_stream,
    merge_draw_lists,
    camera_pv,
    cull_all_triangles,
    tlas_instances,
    light_draw_lists,
    run_skinned,
)
