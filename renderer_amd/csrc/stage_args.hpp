// stage_args.hpp — argument blocks and host-side launchers of every kernel that is built in the library's SECOND
// translation unit (stages_tu.hip): the per-triangle stage (row f-1), the multi-view cull and the shadow-pass lists
// (row f-4), the skinning extension, and the commands-first order of the frame kernel (small launches).
// That unit is compiled with -fno-slp-vectorize. Under plain -O3 the SLP vectoriser packs pairs of independent f32
// operations into v_pk_mul_f32 / v_pk_add_f32 and pays for it with one v_mov_b32 per operand pair (115 moves beside
// 136 packed operations per 64-triangle step in round 2's ISA of the triangle kernel; 136 moves in the multi-view
// kernel); on gfx950 a packed f32 instruction takes twice the issue time of a plain one, so packing buys nothing and
// the moves are pure cost. Measured per kernel (profiles/r03_triangle_no_slp_ab.txt, r03_no_slp_by_kernel.txt): per-
// triangle stage -7 %, skinning -11 .. -16 %, four views -6 %, light lists x16 -4 %, frame kernel 100 k .. 800 k
// -2 .. -4 %; the stores-first frame kernel of large launches (the headline) is the one kernel that is not faster
// that way and stays in api_frame.hip, which sees only this header.
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>

namespace mip {

struct MeshEntry;   // instance_kernel.hpp
struct MeshDraw;
struct KernelArgs;

// ---- row f-1: per-triangle stage (triangle_kernels.hpp) ----
struct TriangleArgs {
  uint32_t* cmds;                 // compacted commands of the instance kernel; indexCount is rewritten
  const uint32_t* count;          // number of commands (device)
  const uint32_t* src_index_offset;
  const float4* model;            // n x mat4 of the same frame
  const float* vertices;          // consolidated positions, packed vec3
  unsigned long long vertex_bytes; // their size (the walk reads them through a bounded descriptor)
  const uint32_t* indices;        // consolidated indices
  uint32_t* out_indices;          // culled index stream (uvec3 out_index_buffer[])
  unsigned long long capacity;    // in indices
  uint32_t first_instance_base;
  uint32_t* error_flag;
  uint32_t* help_counter;         // device word: parts whose survivors a waiting part counted itself (MipTimings.prefix_helps)
  uint32_t* ticket;               // next command to hand out; zeroed by the host before the launch
  uint32_t* final_index_count;    // parts kernel only: where a command's final indexCount goes (see RecompactArgs.index_count)
  uint32_t geometry_finite;       // every position passed to mip_set_geometry was finite
  const uint32_t* index_total;    // the frame's total indexCount before the stage (device), for tri_choice; null = no choice
  uint32_t max_lod_tris;          // largest command the mesh table can produce, in triangles
  // round 5, large frames: commands handed out LARGEST FIRST (sort kernels below) and a device-side choice between this grid and the range kernel's
  const uint32_t* order;          // command numbers by descending size class, or null: list order
  uint32_t* sort_info;            // kSortWords words (layout: triangle_kernels.hpp), zeroed by the host per frame; holds the ticket counter too
  uint32_t first_index_base;      // MipFrame.first_index_base (the length of the triangle stream is taken from the last command)
  uint32_t choice_waves;          // waves of the range kernel's grid: the choice rule's measure of the machine
  uint32_t choice_mode;           // 0 = run; 1 = run iff the frame is the range decomposition's; 2 = iff the wave-per-command one's; 3 = prepare kernel: either part
  uint32_t pull_tickets;          // workgroup-per-command kernel: pull commands from `ticket` instead of a static stride: 0 = stride, else the
                                  // command count from which a ticket is FOUR consecutive commands (65 536; MIP_TUNE_TRI_BATCH_FROM for tests)
  float pv[16];
};

struct RecompactArgs {
  const uint32_t* in_cmds;
  const uint32_t* index_count;  // per command: its indexCount after the triangle stage — or null: it has been written into in_cmds itself.
                                // The parts kernel must NOT rewrite the command in place: a part of command c that starts late (its
                                // successors have long computed its survivors themselves and finished) still reads the command's
                                // ORIGINAL indexCount to find its triangle range (found in round 4 by the fault-injection test).
  const uint32_t* in_count;
  uint32_t* out_cmds;
  uint32_t* out_count;
  uint32_t* zero_words;   // n_zero words (<= 1024) cleared at the end: the stage's counters, for the next frame of the slot
  uint32_t n_zero;
};

struct RecompactWideArgs {
  const uint32_t* in_cmds;
  const uint32_t* index_count;  // as RecompactArgs.index_count
  const uint32_t* in_count;
  uint32_t* out_cmds;
  uint32_t* out_count;
  uint32_t* block_base;   // (round-4 three-launch form) one word per 1024 commands: survivors in the block, then their exclusive prefix
  uint32_t n_blocks;
  unsigned long long* block_status;  // one-launch form: one granule per 1024 commands {tag : 32 | inclusive : 1 | value : 31}
  uint32_t epoch;                    // unique per launch on this frame slot, never 0
  uint32_t* help_counter;            // device word (MipTimings.prefix_helps), or null
  uint32_t* zero_words;              // as RecompactArgs
  uint32_t n_zero;
#ifdef MIP_DEBUG_STAMPS
  uint32_t debug_skip;               // diagnostic build only: every fourth workgroup never publishes (its successors count its commands themselves)
#endif
};

constexpr uint32_t kTriParts = 16;
constexpr uint32_t kTriPartMaxT = 8;  // triangles per thread and part: commands up to 16 * 256 * 8 = 32 768 triangles

struct TrianglePartsArgs {
  TriangleArgs t;
  unsigned long long* part_status;  // [commands][kTriParts] granules {epoch : 32 | survivors : 32}
  uint32_t epoch;                   // unique per launch on this frame slot, never 0
#ifdef MIP_DEBUG_STAMPS
  uint32_t debug_reverse;           // diagnostic build only: work items are taken from the LAST one down (later parts first)
  uint32_t debug_skip_part;         // diagnostic build only: part (value - 1) of every command never publishes (0 = off): its successors count it themselves
#endif
};

// Round 5: the stage as equal RANGES of the frame's triangle stream, one per wave (triangle_kernels.hpp, mip_triangle_cull_ranges_kernel).
// ("chunk" in identifiers — TriangleChunkArgs, chunk_walk, TriangleKernel::chunks, MIP_TUNE_TRI_CHUNKS_FROM — is what the first build of this
// kernel called a range.)
struct TriangleChunkArgs {
  TriangleArgs t;                      // (index_total, max_lod_tris, pull_tickets unused; ticket: zeroed by the host)
  uint32_t* range_first_cmd;           // [ranges_cap]: the first command that owns a slot at or behind the range's first slot
  unsigned long long* range_status;    // [ranges_cap] granules {epoch : 32 | survivors of the range's LAST segment : 32}
  uint32_t ranges_cap;
  uint32_t n_waves;                    // waves of the grid the stage is launched over (= 4 * workgroups)
  uint32_t ticket_slots;               // slots per range of a long stream (a wave's first range dealt statically, the others pulled from the counter)
  uint32_t epoch;                      // unique per launch on this frame slot, never 0
  uint32_t first_index_base;           // MipFrame.first_index_base: firstIndex - base is the running sum the slots are numbered by
#ifdef MIP_DEBUG_STAMPS
  uint32_t debug_reverse;              // diagnostic build only: ranges are dealt from the LAST one down
  uint32_t debug_skip_part;            // diagnostic build only: ranges b with b % 16 == value - 1 never publish (0 = off)
#endif
};

// Layout of TriangleArgs.sort_info (kSortWords words; cleared by the re-compaction at the end of every frame, by the host before the
// first). Size class k = floor(log2(triangles)). Histogram and cursors exist in kSortCopies copies (workgroup b uses copy b % 8): every
// workgroup adds to them with one atomic per class, and ~800 adds on ONE address are ~9 us (a same-address atomic takes ~11 ns).
constexpr uint32_t kSortCopies = 8;
constexpr uint32_t kSortHist = 0;                       // [copies][32] commands per size class
constexpr uint32_t kSortCursor = kSortCopies * 32;      // [copies][32] by class: commands of the class placed so far by the copy's workgroups
constexpr uint32_t kSortStart = 2 * kSortCopies * 32;   // [32] by RANK d (d = 0: the largest class): first position of the class in `order`
constexpr uint32_t kSortTickets = kSortStart + 32;      // [32] by rank: tickets up to and including the class
constexpr uint32_t kSortBatch = kSortTickets + 32;      // [32] by rank: commands per ticket
constexpr uint32_t kSortTicket = kSortBatch + 32;       // the ticket counter of the stage
constexpr uint32_t kSortWords = kSortTicket + 32;
static_assert(kSortWords <= 1024, "the re-compaction clears the block with one workgroup");

// Launchers (defined in stages_tu.hip). Each enqueues one kernel on `stream`; errors surface through hipGetLastError.
void launch_triangle_stage(uint32_t map_blocks, uint32_t blocks, hipStream_t stream, const TriangleChunkArgs& a);  // frames above tri_block_max: range map OR size-class
                                                                                                                  // histogram, scatter, then ONE grid that takes either decomposition
uint32_t triangle_chunks_blocks_per_cu();  // workgroups of the range kernel a CU holds at once: its ranges are dealt over a grid that is resident as a whole
void launch_triangle_cull_chunks(uint32_t map_blocks, uint32_t blocks, hipStream_t stream, const TriangleChunkArgs& a);  // range map, then the stage
void launch_triangle_cull_waves(uint32_t blocks, hipStream_t stream, const TriangleArgs& a);
void launch_triangle_cull_block(uint32_t threads /* 256 | 512 | 1024 */, uint32_t blocks, hipStream_t stream, const TriangleArgs& a);
void launch_triangle_cull_parts(uint32_t blocks, hipStream_t stream, const TrianglePartsArgs& a);
void launch_recompact(hipStream_t stream, const RecompactArgs& a);
void launch_recompact_wide(hipStream_t stream, const RecompactWideArgs& a);  // one launch (block_status set) or count, scan, scatter

// ---- row f-4: shadow-pass draw lists (light_lists_kernel.hpp) ----
constexpr uint32_t kMaxLights = 16;  // the shadow atlas is DIM x DIM = 4 x 4 maps, shadow_mapping.rs:24

struct LightListArgs {
  const float* pos;          // n*3
  const uint32_t* mesh_id;   // n
  const MeshEntry* meshes;   // m
  const MeshDraw* mesh_draw; // m
  uint32_t* out;             // n_lights * n * 5 words
  uint32_t n;
  uint32_t n_lights;
  uint32_t first_instance_base;
  float light[kMaxLights][3];
};

void launch_light_draw_lists(bool aligned16, uint32_t tiles, hipStream_t stream, const LightListArgs& a);

// ---- extension, BASELINE config 5: skinned instances (skinning_kernel.hpp) ----
constexpr uint32_t kMaxJoints = 32;
constexpr uint32_t kPoseWords = 10;  // t xyz, q ijkw, s xyz
constexpr uint32_t kSkinBlock = 256;

struct alignas(16) JointEntry {
  float ibm[12];     // rows 0..2 of inverseBindMatrices[k], column-major 3x4
  float box[6];      // min xyz, max xyz of the bind-pose vertices weighted to this joint; min > max: none
  int32_t parent;    // < k, or -1
  uint32_t sorted;   // entry i: the i-th joint in depth order and its parent, joint | parent << 8
};
static_assert(sizeof(JointEntry) == 80, "JointEntry layout");

struct SkinArgs {
  const float* poses;          // n * J * 10, 8-byte aligned
  const JointEntry* joints;    // J
  float4* palette;             // n * J * 4 (mat4 column-major) or null
  float* local_box;            // n*8: {min xyz, -, max xyz, -} of the posed mesh (the fold's raw result; slots 3 and 7 unused)
  uint32_t n;
  uint32_t n_joints;
  uint32_t max_depth;
  uint32_t inv_joints;                      // ceil(2^16 / J): x / J == (x * inv) >> 16 for x < 256
  float box_bound;                          // 3 * max |joint box coordinate| + 1 over the skeleton; +inf: some box is not finite
  uint32_t level_inv[kMaxJoints + 1];       // ceil(2^16 / joints at depth d)
  uint8_t level_start[kMaxJoints + 2];      // depth d owns sorted entries [level_start[d], level_start[d+1])
};

void launch_skinned_bounds(uint32_t blocks, hipStream_t stream, const SkinArgs& a);

// ---- row f-4: up to four culled views per launch (views_kernel.hpp) ----
constexpr uint32_t kMaxViews = 4;

struct ViewArgs {
  static constexpr bool kFirstMoverAdds = false;  // only owners add to a view's group accumulators (KernelArgs has the other rule)
  // prefix state of this view (publish_aggregate / resolve_prefix read these names)
  unsigned long long* status0;
  unsigned long long* acc1;
  unsigned long long* start1;
  uint32_t groups_cap;
  uint32_t group_shift;
  uint32_t epoch;
  uint32_t* error_flag;
  uint32_t* help_counter;  // device word (MipTimings.prefix_helps)
#ifdef MIP_DEBUG_STAMPS
  unsigned long long* stamps;  // never set: keeps the shared prefix routines compiling in the diagnostic build
#endif
  // outputs
  uint32_t* bitmap;       // ceil(n/32) words or null
  uint32_t* cmds;         // n*5 words
  uint32_t* draw_count;
  uint32_t* index_total;  // or null
  uint32_t first_instance_base;
  uint32_t first_index_base;
  float planes[24];
  float cam[3];
};

struct ViewsArgs {
  const float* pos;
  const float4* rot;
  const float* scale;
  const uint32_t* mesh_id;
  const MeshEntry* meshes;
  const MeshDraw* mesh_draw;
  uint32_t n;
  uint32_t n_tiles;
  uint32_t bitmap_words;
  uint32_t n_views;
#ifdef MIP_DEBUG_STAMPS
  uint32_t debug_tile_mult, debug_tile_add;  // diagnostic build only: tile = (blockIdx.x * mult + add) % n_tiles (KernelArgs has the same)
#endif
  ViewArgs view[kMaxViews];
};

void launch_cull_views(bool general, uint32_t tiles, hipStream_t stream, const ViewsArgs& a);

// ---- rows a-1 .. a-7, commands-first order (kOrder == 3 of instance_kernel.hpp: launches below ~0.9 M instances) ----
// The kernel to hand to hipLaunchKernel; the stores-first order is instantiated in api_frame.hip.
using FrameKernelFn = void (*)(const float*, const float4*, const float*, const uint32_t*, const MeshEntry*, uint32_t*, uint32_t, uint32_t, const KernelArgs);
// kernelParams of a launch of the frame kernel: its leading scalars are copies of the block's fields (instance_kernel.hpp, MIP_FRAME_HEAD_PARAMS)
struct FrameKernelParams {
  void* p[9];
  explicit FrameKernelParams(KernelArgs& a) : p{&a.pos, &a.rot, &a.scale, &a.mesh_id, &a.meshes, &a.cmds, &a.n, &a.one_mesh, &a} {}
};
FrameKernelFn frame_kernel_commands_first(bool box_override, bool general, int wire /* 0 | 1 | 2 = packed */, bool first_mover);

}  // namespace mip
