"""TEST-ONLY stand-in for renderer_amd.InstancePipeline backed by the CPU oracle, so the
host-side shard/exchange logic (renderer_amd/sharded.py) can run under gloo without a GPU.
"Device pointers" are addresses of CPU torch tensors. Never imported by the product."""
import ctypes as C

import numpy as np

import oracle
from renderer_amd.pipeline import DRAW_CMD_DTYPE, SHARD_HEADER_BYTES


def _view(ptr, nbytes, dtype):
    buf = (C.c_char * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


class OraclePipeline:
    def __init__(self, scene):
        self.s = scene
        self.n = scene["n"]

    def run_device(self, frame, model=0, visible_bitmap=0, draw_cmds=0, draw_count=0, draw_index_total=0,
                   world_aabb=0, async_=False):
        s = self.s
        r = oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], np.array(frame.planes[:], np.float32),
                       np.array(frame.cam_pos[:], np.float32), first_instance_base=frame.first_instance_base,
                       first_index_base=frame.first_index_base)
        n = self.n
        if model:
            _view(model, n * 64, np.float32)[:] = r["model"].reshape(-1)
        if visible_bitmap:
            _view(visible_bitmap, ((n + 31) // 32) * 4, np.uint32)[:] = r["visible_bitmap"]
        if draw_cmds:
            c = r["draw_count"]
            _view(draw_cmds, c * 20, np.uint8)[:] = r["draw_cmds"].view(np.uint8).reshape(-1)
            _view(draw_count, 4, np.uint32)[0] = c
            if draw_index_total:
                _view(draw_index_total, 4, np.uint32)[0] = r["draw_index_total"]

    def wait(self):
        """As mip_wait: reports a deferred MIP_ERR_CAPACITY of an (async) merge."""
        if getattr(self, "_overflow", False):
            self._overflow = False
            from renderer_amd._lib import MipError

            raise MipError(-4, "a shard's draw list is longer than the exchanged chunk holds; merged list truncated")

    def merge_draw_lists(self, chunks_ptr, n_chunks, stride, out_cmds_ptr, out_count_ptr, async_=False, chunk_capacity=0):
        lists, totals = [], []
        fits = (stride - SHARD_HEADER_BYTES) // 20
        capacity = min(chunk_capacity, fits) if chunk_capacity else fits
        for k in range(n_chunks):
            h = _view(chunks_ptr + k * stride, 8, np.uint32)
            count = int(h[0])
            if count > capacity:  # what mip_merge_draw_lists_kernel does: cut there and raise the flag
                count = capacity
                self._overflow = True
            lists.append(_view(chunks_ptr + k * stride + SHARD_HEADER_BYTES, count * 20, np.uint8).view(DRAW_CMD_DTYPE).copy())
            totals.append(int(h[1]))
        merged, index_total = oracle.merge_draw_lists(lists, totals)
        _view(out_cmds_ptr, len(merged) * 20, np.uint8)[:] = merged.view(np.uint8).reshape(-1)
        oc = _view(out_count_ptr, 8, np.uint32)
        oc[0] = len(merged)
        oc[1] = index_total
