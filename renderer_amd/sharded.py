"""Sharded scenes: one process per GPU, contiguous draw_index shards, and ONE all-gather of
the compacted draw lists (torch.distributed: backend "nccl" = RCCL over xGMI on MI355X,
"gloo" in the CPU tests) followed by a merge that concatenates the shards in rank order and
rebases firstIndex.

Every rank contributes one fixed-size chunk [MipShardHeader | the list in its WIRE form: records
in blocks of 256 behind a 16-byte block header — packed 4-byte records {instance index | mesh <<
index_bits | lod << 31}, 4.06 B per command instead of 20, whenever the largest shard fits the index
bits the mesh ids leave (MIP_OUT_WIRE_PACKED), else 8-byte records {firstInstance, mesh | lod << 31}
(MIP_OUT_WIRE) — include/mi_instance_pipeline.h]; the pipeline kernel
writes count / index total / records straight into that chunk and the merge kernel expands the
records against the replicated mesh table, so a frame is kernel -> all_gather_into_tensor ->
merge kernel with no host round trip. (`wire=False` exchanges the 20-byte commands themselves:
the round-2 format, kept for A/B runs; the merged list is byte-identical either way.) The capacity
defaults to the shard size and can be tightened from the counts a previous frame produced
(`tighten`). A frame that overflows a tightened chunk (the camera moved) is not lost: every rank
sees the same gathered headers, so every rank's merge reports the overflow, and `complete()`
repeats that frame's all-gather + merge once at full capacity — this rank's complete list is
still in its send buffer — before handing the result out.

No multi-GPU scaling curve exists for this path yet: it has run with world size 1 on an MI355X and
with world sizes 2 and 3 under gloo on CPU; N = 2/4/8 numbers come from the round-end driver.
"""
import numpy as np

from ._lib import MipError
from .pipeline import SHARD_HEADER_BYTES, make_frame, wire_body_bytes, wire_form, wire_index_bits

MIP_ERR_CAPACITY = -4

CMD_BYTES = 20
ALLGATHER_MIN_INSTANCES = 1_000_000  # north star: exchange the draw list only at >= 1 M instances


def shard_range(n_global, world, rank):
    """Contiguous draw_index range of `rank`: ceil(N/R) per rank, the last ones possibly short/empty."""
    per = (n_global + world - 1) // world
    lo = min(n_global, rank * per)
    hi = min(n_global, lo + per)
    return lo, hi


def chunk_stride_bytes(capacity, wire=False):
    """wire: False = 20-byte commands, True / 1 = 8-byte wire records, "packed" / 2 = packed 4-byte records."""
    form = wire_form(wire)
    stride = SHARD_HEADER_BYTES + (wire_body_bytes(capacity, packed=form == 2) if form else capacity * CMD_BYTES)
    return (stride + 255) // 256 * 256


class DrawListExchange:
    """Frame driver for one rank of a sharded scene.

    `pipe` needs run_device(frame, **ptrs) and merge_draw_lists(...) with the semantics of
    renderer_amd.InstancePipeline (the HIP context in the product; the tests inject a
    CPU stand-in so the exchange logic runs under gloo)."""

    def __init__(self, pipe, n_local, world, rank, device, dist=None, torch=None, group=None, capacity=None, wire=True,
                 n_meshes=None):
        """wire: True = the wire form, PACKED when the largest shard fits the index bits a table of `n_meshes` entries
        leaves (n_meshes defaults to pipe.n_meshes; unknown: 8-byte records); 1 = 8-byte records; "packed" = packed or
        an error; False = 20-byte commands. Every rank must pass the same value — and holds the same mesh table."""
        if torch is None:
            import torch
        if dist is None:
            import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.pipe, self.n_local, self.world, self.rank, self.device = pipe, int(n_local), int(world), int(rank), device
        self._on_gpu = getattr(torch.device(device), "type", "cpu") == "cuda"
        self.wire = bool(wire)
        # Chunk sizes must be the same on every rank (an all-gather of unequal pieces is a collective mismatch),
        # but the last shard of ceil(N/R)-sized ranges may be shorter: everything is sized by the LARGEST shard.
        # One tiny all-reduce at construction (collective: every rank constructs its exchange at the same point).
        nmax = torch.tensor([self.n_local], dtype=torch.int64, device=device)
        if self.world > 1:
            dist.all_reduce(nmax, op=dist.ReduceOp.MAX, group=group)
        self.n_max = int(nmax.item())
        # the form of the list: derived from numbers every rank has (the largest shard, the replicated table's size)
        if n_meshes is None:
            n_meshes = getattr(pipe, "n_meshes", None)
        fits = n_meshes is not None and self.n_max <= (1 << wire_index_bits(n_meshes))
        if wire is True:
            self.form = 2 if fits else 1
        else:
            self.form = wire_form(wire)
            if self.form == 2 and not fits:
                raise ValueError(f"packed wire records: shards of up to {self.n_max} instances do not fit beside {n_meshes} mesh ids")
        # the kernel may emit up to n_local commands, so the send buffer always has room for all of them
        self._send_full = torch.zeros(chunk_stride_bytes(self.n_max, self.form) // 4, dtype=torch.int32, device=device)
        self.merged_count = torch.zeros(2, dtype=torch.int32, device=device)
        self.retries = 0       # frames re-gathered at full capacity after a tightened chunk overflowed
        self._in_flight = 0    # frames issued since the last complete()
        self.set_capacity(self.n_max if capacity is None else capacity)

    def set_capacity(self, capacity):
        torch = self.torch
        self.capacity = int(min(max(capacity, 0), self.n_max))  # the same number on every rank
        self.stride = chunk_stride_bytes(self.capacity, self.form)
        words = self.stride // 4
        self.send = self._send_full[:words]
        self.recv = torch.empty(self.world * words, dtype=torch.int32, device=self.device)
        self.merged = torch.empty((max(self.world * self.capacity, 1), 5), dtype=torch.int32, device=self.device)

    def _check_stream(self, what):
        # kernel -> all-gather -> merge are ordered by ONE stream: the collective goes to torch's current stream, so a
        # HIP context must have been created on that very stream (the CPU stand-ins of the tests have no `stream`)
        if self._on_gpu and hasattr(self.pipe, "stream"):
            current = self.torch.cuda.current_stream(self.device).cuda_stream
            if self.pipe.stream is None or int(self.pipe.stream) != int(current):
                raise ValueError(f"DrawListExchange.{what}: the pipeline's stream is not torch's current stream — create it with "
                                 "InstancePipeline(..., stream=torch.cuda.current_stream(device).cuda_stream) and call it "
                                 "under that stream; the all-gather would otherwise race with the kernels")

    def step(self, frame, outs=None, model=0, visible_bitmap=0, world_aabb=0, kernel_done=None):
        """One frame on this rank. `outs` (optional) supplies model / bitmap device buffers. `kernel_done`
        (optional, a torch.cuda.Event) is recorded right behind the shard kernel, in front of the all-gather."""
        if outs is not None:
            model = outs.model.data_ptr()
            visible_bitmap = outs.bitmap.data_ptr()
        self._check_stream("step")
        base = self._send_full.data_ptr()
        self.pipe.run_device(frame, model=model, visible_bitmap=visible_bitmap, world_aabb=world_aabb,
                             draw_cmds=base + SHARD_HEADER_BYTES, draw_count=base, draw_index_total=base + 4,
                             async_=True, **({"wire": self.form} if self.form else {}))
        if kernel_done is not None:
            kernel_done.record()
        self._gather_and_merge()
        self._in_flight += 1

    def _gather_and_merge(self):
        self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        if self.form:
            self.pipe.merge_wire_lists(self.recv.data_ptr(), self.world, self.stride, self.merged.data_ptr(), self.merged_count.data_ptr(),
                                       async_=True, chunk_capacity=self.capacity, **({"packed": True} if self.form == 2 else {}))
        else:
            self.pipe.merge_draw_lists(self.recv.data_ptr(), self.world, self.stride, self.merged.data_ptr(), self.merged_count.data_ptr(),
                                       async_=True, chunk_capacity=self.capacity)

    def complete(self):
        """Block until the frames issued so far are done. If the LAST one overflowed its tightened chunk,
        repeat its all-gather + merge at full capacity (collective: the overflow is visible in the headers
        every rank gathered, so every rank takes this branch together). Returns True if it had to."""
        if self._in_flight:
            self._check_stream("complete")  # a repair issues an all-gather: it must land on the pipeline's stream
        in_flight, self._in_flight = self._in_flight, 0
        err = None
        try:
            self.pipe.wait()
        except MipError as e:
            err = e
        if err is None:
            return False
        # Whether to repair is decided from data EVERY rank holds — the gathered headers — never from which error code this
        # rank happened to get: the merge kernel raises the overflow on every rank alike, but a rank with a second, local
        # error (a corrupt record, an external semaphore that expired) is handed THAT code by mip_wait, and a rank that then
        # skipped the repair's all-gather would leave its peers blocked in it for good (round-3 advisor finding).
        overflow = False
        if in_flight:
            # the gathered headers are read back through the device: after a FATAL device error that read may itself fail or
            # return garbage (a count above every shard's size) — then there is nothing to repair, and the error that counts is
            # the one mip_wait reported (round-4 advisor finding)
            try:
                counts, _ = self.counts()
                largest = int(counts.max())
            except Exception:  # noqa: BLE001 (torch raises RuntimeError subclasses on a dead device)
                raise err from None
            overflow = self.capacity < largest <= self.n_max
        if not overflow:
            raise err
        if in_flight != 1:
            raise MipError(MIP_ERR_CAPACITY, "a tightened chunk overflowed with several frames in flight: the overflowing "
                                             "frame's list has been overwritten; call complete() after every frame") from err
        self.set_capacity(self.n_max)
        self._gather_and_merge()
        self.pipe.wait()
        self.retries += 1
        if err.code != MIP_ERR_CAPACITY:
            raise err  # this rank's own error, reported AFTER it has taken part in the collective repair
        return True

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

        self.complete()
        total, index_total = (int(x) & 0xFFFFFFFF for x in self.merged_count.cpu().tolist())
        cmds = self.merged[:total].cpu().numpy().view(np.uint32).reshape(-1).view(DRAW_CMD_DTYPE)
        return cmds.copy(), total, index_total


class PipelinedExchange:
    """F frames in flight on the sharded path: F contexts, each bound to its own torch stream
    (kernel -> all-gather -> merge stay ordered within a frame), issued round-robin so that
    frame k+1's kernel runs while frame k's draw lists are still on the wire.

    What overlaps is a shard KERNEL with the previous frames' all-gather and merge — never two shard kernels:
    slot k's kernel waits (stream-side, an event) for the previous slot's kernel. Two spin-waiting launches that
    are each only partly resident are the shape that deadlocked in round 2 (every resident
    tile of one waits for a tile that cannot start because the other's waiting tiles hold the CUs, and vice
    versa); a collective kernel beside a shard kernel is not that shape — its workgroups wait for peers, never
    for this GPU's shard kernel, so they always drain (tests/fake_ccl/spin_rccl.hip rehearses exactly this)."""

    def __init__(self, make_pipe, n_local, world, rank, device, frames=2, dist=None, torch=None, group=None, wire=True, n_meshes=None):
        if torch is None:
            import torch
        self.torch = torch
        self.on_gpu = getattr(device, "type", str(device)) == "cuda"
        # on a GPU every frame slot has its own stream; the CPU tests (gloo) run the same rotation on the host
        self.streams = [torch.cuda.Stream(device=device) if self.on_gpu else None for _ in range(frames)]
        self.pipes = [make_pipe(st.cuda_stream if st is not None else 0) for st in self.streams]
        self.exchanges = [DrawListExchange(p, n_local, world, rank, device, dist=dist, torch=torch, group=group, wire=wire, n_meshes=n_meshes)
                          for p in self.pipes]
        self.kernel_done = [torch.cuda.Event() if self.on_gpu else None for _ in range(frames)]
        self.last = None  # slot whose shard kernel was issued last
        self.next = 0

    def _under(self, k, fn):
        """Runs fn under slot k's stream: everything an exchange enqueues (also the repair of an overflowed chunk,
        whose all-gather goes to torch's CURRENT stream) must be ordered with that slot's kernels and merges."""
        if self.on_gpu:
            with self.torch.cuda.stream(self.streams[k]):
                return fn()
        return fn()

    def step(self, frame, outs_per_frame):
        """Issues one frame on the next slot. A slot is reused every `frames` steps: its previous frame is
        completed first (and repaired if its tightened chunk overflowed), so no list is ever overwritten
        before it has been handed out."""
        k = self.next
        self.next = (k + 1) % len(self.exchanges)
        ex = self.exchanges[k]

        def issue():
            if ex.capacity < ex.n_max and ex._in_flight:
                ex.complete()
            if self.on_gpu and self.last is not None and self.last != k:
                self.streams[k].wait_event(self.kernel_done[self.last])  # shard kernels never overlap each other
            ex.step(frame, outs_per_frame[k], kernel_done=self.kernel_done[k])

        self._under(k, issue)
        self.last = k
        return k

    def wait(self):
        """Drains every frame slot; a slot whose last frame overflowed its tightened chunk is repaired
        (DrawListExchange.complete). Collective, like step()."""
        return [self._under(k, ex.complete) for k, ex in enumerate(self.exchanges)]

    def tighten(self, margin=1.0625):
        return [ex.tighten(margin) for ex in self.exchanges]

    def merged_draw_list(self, k):
        """Slot k's merged list on the host (completes — and if need be repairs — its last frame first)."""
        return self._under(k, self.exchanges[k].merged_draw_list)

    def close(self):
        for p in self.pipes:
            p.close()


def make_shard_frame(planes, cam_pos, n_global, world, rank):
    lo, _ = shard_range(n_global, world, rank)
    return make_frame(planes, cam_pos, first_instance_base=lo)
