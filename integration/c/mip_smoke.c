/* A plain C99 host for libmi_instance_pipeline.so: what the reference-side shim does, minus Rust
 * ("exercised only through a C test driver"). Builds a tiny scene, runs one frame
 * with host output pointers and prints the draw count and a checksum of the command bytes.
 *   gcc -std=c99 -I include integration/c/mip_smoke.c -L renderer_amd/lib -lmi_instance_pipeline -o mip_smoke */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi_instance_pipeline.h"

int main(void) {
  enum { N = 1000 };
  MipConfig cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.max_instances = N;
  cfg.max_meshes = 1;
  MipContext* ctx = NULL;
  int32_t rc = mip_create(&cfg, &ctx);
  if (rc != MIP_OK) {
    printf("mip_create: %d\n", rc);
    return 10 - rc;
  }
  MipMesh mesh;
  memset(&mesh, 0, sizeof mesh);
  for (int a = 0; a < 3; ++a) {
    mesh.aabb_min[a] = -0.5f;
    mesh.aabb_max[a] = 0.5f;
  }
  mesh.n_lods = 2;
  mesh.index_len[0] = 36;
  mesh.index_len[1] = 18;
  mesh.index_offset[1] = 36;
  static float pos[N * 3], rot[N * 4], scale[N];
  static uint32_t mesh_id[N];
  for (int i = 0; i < N; ++i) { /* a 10 x 10 x 10 lattice in front of the default camera */
    pos[3 * i + 0] = (float)(i % 10) * 3.0f - 13.5f;
    pos[3 * i + 1] = (float)((i / 10) % 10) * 3.0f - 13.5f;
    pos[3 * i + 2] = (float)(i / 100) * 6.0f + 4.0f;
    rot[4 * i + 3] = 1.0f;
    scale[i] = 1.0f;
  }
  if ((rc = mip_set_mesh_table(ctx, &mesh, 1)) || (rc = mip_set_instances(ctx, pos, rot, scale, mesh_id, N))) {
    printf("setup: %d %s\n", rc, mip_last_error(ctx));
    return 10 - rc;
  }
  /* the reference's default camera: eye (0,1,2), looking down +z, fov 70 deg, aspect 2, near 0.1, far 100;
   * planes as -(row3 +- row k) of projection * view, order L R B T N F (src/ecs.rs:78-90) */
  MipFrame frame;
  memset(&frame, 0, sizeof frame);
  const float t = 0.70020754f /* tan(35 deg) */, aspect = 2.0f, n = 0.1f, f = 100.0f;
  const float px = 1.0f / (aspect * t), py = 1.0f / t, pz = f / (f - n), pw = -(f * n) / (f - n);
  const float eye[3] = {0.0f, 1.0f, 2.0f};
  /* rows of projection * view with view = translate(-eye) */
  const float r0[4] = {px, 0, 0, -px * eye[0]}, r1[4] = {0, py, 0, -py * eye[1]}, r2[4] = {0, 0, pz, -pz * eye[2] + pw},
              r3[4] = {0, 0, 1, -eye[2]};
  const float* rows[3] = {r0, r1, r2};
  for (int k = 0; k < 3; ++k)
    for (int c = 0; c < 4; ++c) {
      frame.planes[(2 * k) * 4 + c] = -(r3[c] + rows[k][c]);
      frame.planes[(2 * k + 1) * 4 + c] = -(r3[c] - rows[k][c]);
    }
  memcpy(frame.cam_pos, eye, sizeof eye);
  static float model[N * 16];
  static uint32_t bitmap[(N + 31) / 32];
  static MipDrawIndexedIndirectCommand cmds[N];
  uint32_t count = 0, index_total = 0;
  MipOutputs out;
  memset(&out, 0, sizeof out);
  out.model = model;
  out.visible_bitmap = bitmap;
  out.draw_cmds = cmds;
  out.draw_count = &count;
  out.draw_index_total = &index_total;
  out.flags = MIP_OUT_HOST;
  if ((rc = mip_run(ctx, &frame, &out))) {
    printf("mip_run: %d %s\n", rc, mip_last_error(ctx));
    return 10 - rc;
  }
  uint32_t visible = 0, sum = 0;
  for (uint32_t w = 0; w < (N + 31) / 32; ++w)
    for (uint32_t b = bitmap[w]; b; b &= b - 1) ++visible;
  for (uint32_t k = 0; k < count; ++k) sum = sum * 31u + cmds[k].indexCount + cmds[k].firstIndex * 7u + cmds[k].firstInstance * 13u;
  /* the running firstIndex must be the sum of the earlier index counts */
  uint32_t running = 0;
  int ok = count == visible;
  for (uint32_t k = 0; k < count; ++k) {
    ok = ok && cmds[k].firstIndex == running && cmds[k].instanceCount == 1 && (cmds[k].indexCount == 36 || cmds[k].indexCount == 18);
    running += cmds[k].indexCount;
  }
  ok = ok && running == index_total && model[12] == pos[0] && model[15] == 1.0f;
  printf("C_SMOKE %s instances=%d visible=%u commands=%u index_total=%u checksum=%08x\n", ok ? "OK" : "BAD", N, visible, count,
         index_total, sum);
  mip_destroy(ctx);
  mip_destroy(NULL); /* idempotent on NULL */
  return ok ? 0 : 1;
}
