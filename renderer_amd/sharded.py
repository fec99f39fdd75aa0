"""Sharded scenes: one process per GPU, contiguous draw_index shards, and ONE all-gather of
the compacted draw lists (torch.distributed: backend "nccl" = RCCL over xGMI on MI355X,
"gloo" in the CPU tests) followed by a merge that concatenates the shards in rank order and
rebases firstIndex.

Every rank contributes one fixed-size chunk [MipShardHeader | capacity x 20-B commands]; the
pipeline kernel writes count / index total / commands straight into that chunk, so a frame is
kernel -> all_gather_into_tensor -> merge kernel with no host round trip. The capacity
defaults to the shard size and can be tightened from the counts a previous frame produced
(`tighten`); a frame that overflows it is reported (MIP_ERR_CAPACITY), never silently cut.
"""
import numpy as np

from .pipeline import SHARD_HEADER_BYTES, make_frame

CMD_BYTES = 20
ALLGATHER_MIN_INSTANCES = 1_000_000  # north star: exchange the draw list only at >= 1 M instances


def shard_range(n_global, world, rank):
    """Contiguous draw_index range of `rank`: ceil(N/R) per rank, the last ones possibly short/empty."""
    per = (n_global + world - 1) // world
    lo = min(n_global, rank * per)
    hi = min(n_global, lo + per)
    return lo, hi


def chunk_stride_bytes(capacity):
    stride = SHARD_HEADER_BYTES + capacity * CMD_BYTES
    return (stride + 255) // 256 * 256


class DrawListExchange:
    """Frame driver for one rank of a sharded scene.

    `pipe` needs run_device(frame, **ptrs) and merge_draw_lists(...) with the semantics of
    renderer_amd.InstancePipeline (the HIP context in the product; the tests inject a
    CPU stand-in so the exchange logic runs under gloo)."""

    def __init__(self, pipe, n_local, world, rank, device, dist=None, torch=None, group=None, capacity=None):
        if torch is None:
            import torch
        if dist is None:
            import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.pipe, self.n_local, self.world, self.rank, self.device = pipe, int(n_local), int(world), int(rank), device
        # the kernel may emit up to n_local commands, so the send buffer always has room for all of them
        self._send_full = torch.zeros(chunk_stride_bytes(self.n_local) // 4, dtype=torch.int32, device=device)
        self.merged_count = torch.zeros(2, dtype=torch.int32, device=device)
        self.set_capacity(self.n_local if capacity is None else capacity)

    def set_capacity(self, capacity):
        torch = self.torch
        self.capacity = int(min(max(capacity, 0), self.n_local))
        self.stride = chunk_stride_bytes(self.capacity)
        words = self.stride // 4
        self.send = self._send_full[:words]
        self.recv = torch.empty(self.world * words, dtype=torch.int32, device=self.device)
        self.merged = torch.empty((max(self.world * self.capacity, 1), 5), dtype=torch.int32, device=self.device)

    def step(self, frame, outs=None, model=0, visible_bitmap=0, world_aabb=0):
        """One frame on this rank. `outs` (optional) supplies model / bitmap device buffers."""
        if outs is not None:
            model = outs.model.data_ptr()
            visible_bitmap = outs.bitmap.data_ptr()
        base = self._send_full.data_ptr()
        self.pipe.run_device(frame, model=model, visible_bitmap=visible_bitmap, world_aabb=world_aabb,
                             draw_cmds=base + SHARD_HEADER_BYTES, draw_count=base, draw_index_total=base + 4,
                             async_=True)
        self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        self.pipe.merge_draw_lists(self.recv.data_ptr(), self.world, self.stride, self.merged.data_ptr(),
                                   self.merged_count.data_ptr(), async_=True)

    # -- host-side views (synchronising) --
    def local_count(self):
        return int(self._send_full[0].item())

    def counts(self):
        """Per-rank (draw_count, draw_index_total) of the last frame, from the gathered headers."""
        words = self.stride // 4
        h = self.recv.view(self.world, words)[:, :2].cpu().numpy().view(np.uint32)
        return h[:, 0].copy(), h[:, 1].copy()

    def tighten(self, margin=1.0625):
        """Shrink the exchanged chunk to what the last frame needed (max over ranks) plus a margin."""
        counts, _ = self.counts()
        cap = int(np.ceil(int(counts.max()) * margin / 256.0) * 256)
        self.set_capacity(max(cap, 256))
        return self.capacity

    def merged_draw_list(self):
        from .pipeline import DRAW_CMD_DTYPE

        total, index_total = (int(x) & 0xFFFFFFFF for x in self.merged_count.cpu().tolist())
        cmds = self.merged[:total].cpu().numpy().view(np.uint32).reshape(-1).view(DRAW_CMD_DTYPE)
        return cmds.copy(), total, index_total


class PipelinedExchange:
    """F frames in flight on the sharded path: F contexts, each bound to its own torch stream
    (kernel -> all-gather -> merge stay ordered within a frame), issued round-robin so that
    frame k+1's kernel runs while frame k's draw lists are still on the wire."""

    def __init__(self, make_pipe, n_local, world, rank, device, frames=2, dist=None, torch=None, group=None):
        if torch is None:
            import torch
        self.torch = torch
        self.streams = [torch.cuda.Stream(device=device) for _ in range(frames)]
        self.pipes = [make_pipe(st.cuda_stream) for st in self.streams]
        self.exchanges = [DrawListExchange(p, n_local, world, rank, device, dist=dist, torch=torch, group=group)
                          for p in self.pipes]
        self.next = 0

    def step(self, frame, outs_per_frame):
        k = self.next
        self.next = (k + 1) % len(self.exchanges)
        with self.torch.cuda.stream(self.streams[k]):
            self.exchanges[k].step(frame, outs_per_frame[k])
        return k

    def wait(self):
        for p in self.pipes:
            p.wait()

    def tighten(self, margin=1.0625):
        return [ex.tighten(margin) for ex in self.exchanges]

    def close(self):
        for p in self.pipes:
            p.close()


def make_shard_frame(planes, cam_pos, n_global, world, rank):
    lo, _ = shard_range(n_global, world, rank)
    return make_frame(planes, cam_pos, first_instance_base=lo)
