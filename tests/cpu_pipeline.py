"""TEST-ONLY stand-in for renderer_amd.InstancePipeline backed by the CPU oracle, so the
host-side shard/exchange logic (renderer_amd/sharded.py) can run under gloo without a GPU.
"Device pointers" are addresses of CPU torch tensors. Never imported by the product."""
import ctypes as C

import numpy as np

import oracle
from renderer_amd.pipeline import DRAW_CMD_DTYPE, SHARD_HEADER_BYTES, wire_body_bytes, wire_form, wire_index_bits

WIRE_BLOCK = 256            # MIP_WIRE_BLOCK_COMMANDS
WIRE_SUB = 64               # MIP_WIRE_SUB_BLOCK_COMMANDS: every 64 records carry their own firstIndex anchor
WIRE_BLOCK_WORDS = 4 + 2 * WIRE_BLOCK
WIRE_PACKED_BLOCK = 64      # MIP_WIRE_PACKED_BLOCK_COMMANDS
WIRE_PACKED_BLOCK_WORDS = 4 + WIRE_PACKED_BLOCK   # MIP_OUT_WIRE_PACKED: one word per record


def encode_wire(cmds, mesh_of_cmd, far_of_cmd):
    """The wire form of a 20-byte command list (include/mi_instance_pipeline.h, MIP_OUT_WIRE), restated in numpy:
    blocks of 256 records {firstInstance, mesh | far << 31}, each behind a 16-byte header whose word q is the firstIndex of
    the block's record 64 q. Returns the body as uint32 words (whole blocks; unused slots and anchors zero)."""
    n = len(cmds)
    blocks = (n + WIRE_BLOCK - 1) // WIRE_BLOCK
    body = np.zeros(blocks * WIRE_BLOCK_WORDS, np.uint32)
    v = body.reshape(blocks, WIRE_BLOCK_WORDS)
    if n:
        anchors = np.zeros(blocks * 4, np.uint32)
        a = cmds["firstIndex"][::WIRE_SUB]
        anchors[: len(a)] = a
        v[:, :4] = anchors.reshape(blocks, 4)
        rec = np.zeros((blocks * WIRE_BLOCK, 2), np.uint32)
        rec[:n, 0] = cmds["firstInstance"]
        rec[:n, 1] = np.asarray(mesh_of_cmd, np.uint32) | (np.asarray(far_of_cmd, np.uint32) << np.uint32(31))
        v[:, 4:] = rec.reshape(blocks, 2 * WIRE_BLOCK)
    return body


def wire_live_mask(count, packed=False):
    """Which words of a wire body of `count` records are specified (anchors of sub-blocks that exist, live records)."""
    if packed:
        blocks = (count + WIRE_PACKED_BLOCK - 1) // WIRE_PACKED_BLOCK
        live = np.zeros((blocks, WIRE_PACKED_BLOCK_WORDS), bool)
        live[:, :4] = True
        slots = np.arange(blocks * WIRE_PACKED_BLOCK).reshape(blocks, WIRE_PACKED_BLOCK)
        live[:, 4:] = slots < count
        return live.reshape(-1)
    blocks = (count + WIRE_BLOCK - 1) // WIRE_BLOCK
    live = np.zeros((blocks, WIRE_BLOCK_WORDS), bool)
    subs = np.arange(blocks * 4).reshape(blocks, 4)
    live[:, :4] = subs * WIRE_SUB < count
    slots = np.arange(blocks * WIRE_BLOCK).reshape(blocks, WIRE_BLOCK)
    live[:, 4:] = np.repeat(slots < count, 2, axis=1)
    return live.reshape(-1)


def encode_wire_packed(cmds, mesh_of_cmd, far_of_cmd, first_instance_base, n_meshes):
    """The PACKED wire form (MIP_OUT_WIRE_PACKED), restated in numpy: one word per command, instance index in the frame
    | mesh << index_bits | far << 31, in blocks of 64; block header {firstIndex of the block's first command,
    first_instance_base, index_bits, 0}. Returns the body as uint32 words (whole blocks; unused slots zero)."""
    n = len(cmds)
    bits = wire_index_bits(n_meshes)
    blocks = (n + WIRE_PACKED_BLOCK - 1) // WIRE_PACKED_BLOCK
    body = np.zeros(blocks * WIRE_PACKED_BLOCK_WORDS, np.uint32)
    v = body.reshape(blocks, WIRE_PACKED_BLOCK_WORDS)
    if n:
        v[:, 0] = cmds["firstIndex"][::WIRE_PACKED_BLOCK]
        v[:, 1] = np.uint32(first_instance_base)
        v[:, 2] = bits
        idx = (cmds["firstInstance"] - np.uint32(first_instance_base)).astype(np.uint32)
        assert int(idx.max()) < (1 << bits), "the frame does not fit a packed record"
        rec = np.zeros(blocks * WIRE_PACKED_BLOCK, np.uint32)
        rec[:n] = idx | (np.asarray(mesh_of_cmd, np.uint32) << np.uint32(bits)) | (np.asarray(far_of_cmd, np.uint32) << np.uint32(31))
        v[:, 4:] = rec.reshape(blocks, WIRE_PACKED_BLOCK)
    return body


def unpack_wire(body, count):
    """The packed body of `count` records as the 8-byte form's body (blocks of 256 records {firstInstance, mesh | far << 31},
    four anchors per block)."""
    pblocks = (count + WIRE_PACKED_BLOCK - 1) // WIRE_PACKED_BLOCK
    v = np.asarray(body[: pblocks * WIRE_PACKED_BLOCK_WORDS], np.uint32).reshape(pblocks, WIRE_PACKED_BLOCK_WORDS)
    blocks = (count + WIRE_BLOCK - 1) // WIRE_BLOCK
    out = np.zeros((blocks, WIRE_BLOCK_WORDS), np.uint32)
    anchors = np.zeros(blocks * 4, np.uint32)
    anchors[:pblocks] = v[:, 0]
    out[:, :4] = anchors.reshape(blocks, 4)
    r = v[:, 4:]
    bits = v[:, 2:3]
    low = r & np.uint32(0x7FFFFFFF)
    rec = np.zeros((blocks * WIRE_BLOCK, 2), np.uint32)
    rec[: pblocks * WIRE_PACKED_BLOCK, 0] = (v[:, 1:2] + (low & ((np.uint32(1) << bits) - np.uint32(1)))).reshape(-1)
    rec[: pblocks * WIRE_PACKED_BLOCK, 1] = ((low >> bits) | (r & np.uint32(0x80000000))).reshape(-1)
    out[:, 4:] = rec.reshape(blocks, 2 * WIRE_BLOCK)
    return out.reshape(-1)


def decode_wire(body, count, meshes):
    """Expands `count` wire records against the mesh table: what mip_merge_wire_lists_kernel does for one chunk
    (without the rebasing over chunks)."""
    blocks = (count + WIRE_BLOCK - 1) // WIRE_BLOCK
    v = np.asarray(body[: blocks * WIRE_BLOCK_WORDS], np.uint32).reshape(blocks, WIRE_BLOCK_WORDS)
    rec = v[:, 4:].reshape(-1, 2)[:count]
    mesh = rec[:, 1] & np.uint32(0x7FFFFFFF)
    far = rec[:, 1] >> np.uint32(31)
    n_lods = meshes["n_lods"][mesh]
    lens = np.where((far == 1) & (n_lods > 1), meshes["index_len"][mesh, 1], meshes["index_len"][mesh, 0]).astype(np.uint32)
    out = np.zeros(count, DRAW_CMD_DTYPE)
    out["indexCount"] = lens
    out["instanceCount"] = 1
    out["vertexOffset"] = meshes["vertex_offset"][mesh]
    out["firstInstance"] = rec[:, 0]
    anchors = v[:, :4].reshape(-1)
    for q in range((count + WIRE_SUB - 1) // WIRE_SUB):
        sl = slice(q * WIRE_SUB, min(count, (q + 1) * WIRE_SUB))
        l = lens[sl].astype(np.uint64)
        out["firstIndex"][sl] = ((np.cumsum(l) - l + np.uint64(anchors[q])) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return out


def _view(ptr, nbytes, dtype):
    buf = (C.c_char * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


class OraclePipeline:
    def __init__(self, scene):
        self.s = scene
        self.n = scene["n"]
        self.n_meshes = len(scene["meshes"])

    def run_device(self, frame, model=0, visible_bitmap=0, draw_cmds=0, draw_count=0, draw_index_total=0,
                   world_aabb=0, async_=False, wire=False):
        s = self.s
        r = oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], np.array(frame.planes[:], np.float32),
                       np.array(frame.cam_pos[:], np.float32), first_instance_base=frame.first_instance_base,
                       first_index_base=frame.first_index_base)
        n = self.n
        if model:
            _view(model, n * 64, np.float32)[:] = r["model"].reshape(-1)
        if visible_bitmap:
            _view(visible_bitmap, ((n + 31) // 32) * 4, np.uint32)[:] = r["visible_bitmap"]
        if draw_cmds and wire:
            c = r["draw_count"]
            inst = (r["draw_cmds"]["firstInstance"] - np.uint32(frame.first_instance_base)).astype(np.int64)
            cam = np.array(frame.cam_pos[:], np.float32)
            far = np.array([oracle.pick_lod(2, cam, s["pos"][i]) for i in inst], np.uint32)
            if wire_form(wire) == 2:
                body = encode_wire_packed(r["draw_cmds"], s["mesh_id"][inst], far, frame.first_instance_base, len(s["meshes"]))
            else:
                body = encode_wire(r["draw_cmds"], s["mesh_id"][inst], far)
            _view(draw_cmds, body.nbytes, np.uint32)[:] = body
            _view(draw_count, 4, np.uint32)[0] = c
            if draw_index_total:
                _view(draw_index_total, 4, np.uint32)[0] = r["draw_index_total"]
        elif draw_cmds:
            c = r["draw_count"]
            _view(draw_cmds, c * 20, np.uint8)[:] = r["draw_cmds"].view(np.uint8).reshape(-1)
            _view(draw_count, 4, np.uint32)[0] = c
            if draw_index_total:
                _view(draw_index_total, 4, np.uint32)[0] = r["draw_index_total"]

    def wait(self):
        """As mip_wait: reports a deferred MIP_ERR_CAPACITY of an (async) merge — or, when the test has planted one
        (`other_error_once`), ANOTHER error code that mip_wait ranks above the overflow (a corrupt record: MIP_ERR_DEVICE)."""
        if getattr(self, "_overflow", False):
            self._overflow = False
            from renderer_amd._lib import MipError

            code = getattr(self, "other_error_once", None)
            if code is not None:
                self.other_error_once = None
                raise MipError(code, "a local error that outranks the overflow in mip_wait")
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

    def merge_wire_lists(self, chunks_ptr, n_chunks, stride, out_cmds_ptr, out_count_ptr, async_=False, chunk_capacity=0, packed=False):
        lists, totals = [], []
        fits = ((stride - SHARD_HEADER_BYTES) // (WIRE_PACKED_BLOCK_WORDS * 4) * WIRE_PACKED_BLOCK if packed
                else (stride - SHARD_HEADER_BYTES) // (WIRE_BLOCK_WORDS * 4) * WIRE_BLOCK)
        capacity = min(chunk_capacity, fits) if chunk_capacity else fits
        for k in range(n_chunks):
            h = _view(chunks_ptr + k * stride, 8, np.uint32)
            count = int(h[0])
            if count > capacity:
                count = capacity
                self._overflow = True
            body = _view(chunks_ptr + k * stride + SHARD_HEADER_BYTES, wire_body_bytes(count, packed=packed), np.uint32)
            lists.append(decode_wire(unpack_wire(body, count) if packed else body, count, self.s["meshes"]))
            totals.append(int(h[1]))
        merged, index_total = oracle.merge_draw_lists(lists, totals)
        _view(out_cmds_ptr, len(merged) * 20, np.uint8)[:] = merged.view(np.uint8).reshape(-1)
        oc = _view(out_count_ptr, 8, np.uint32)
        oc[0] = len(merged)
        oc[1] = index_total
