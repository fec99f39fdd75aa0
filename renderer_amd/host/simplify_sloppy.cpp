// simplify_sloppy.cpp — the LOD generator of the reference's scene loader, restated.
//
// src/renderer/systems/scene_loader.rs:739-753 builds LODs 1..5 of every primitive with
//     meshopt::simplify::simplify_sloppy_decoder(&indices, &positions, (indices.len() as f32 * 0.5^x) as usize)
// and keeps a result only if `res.len() < indices.len() && !res.is_empty()`. The crate is `meshopt 0.1.9`
// (Cargo.toml:51), a thin wrapper that vendors zeux/meshoptimizer of early 2020 (v0.13) and calls its
// `meshopt_simplifySloppy(destination, indices, index_count, positions, vertex_count, stride, target_index_count)`.
// Neither the crate nor the library is in the reference checkout or in this image, so what follows is a restatement
// of that function's PUBLISHED algorithm (meshoptimizer src/simplifier.cpp, "sloppy" simplifier) from knowledge of
// that source — it could not be diffed against it here, and no fixture of the reference pins it: UNPINNED, like the
// rest of the oracle chain:
//
//   1. positions are rescaled into the unit cube: (p - min) / max extent;
//   2. a uniform grid of g^3 cells quantises them: id = round(x*(g-1)) << 20 | round(y*(g-1)) << 10 | round(z*(g-1));
//      a triangle survives a grid if its three corners fall into three different cells;
//   3. g is searched in (0, 1025) so that the surviving triangle count is the largest one <= target/3: first guess
//      sqrt(target/6) ("~2 triangles per vertex"), then up to five steps of three-point interpolation search, then
//      bisection, 15 passes at most;
//   4. cells are numbered in order of first appearance (by vertex index); every triangle adds the quadric of its
//      plane, weighted by sqrt(area) (x3 when the triangle lies inside one cell), to the cells of its corners;
//   5. each cell is represented by its vertex of least quadric error (the first such vertex on ties);
//   6. every surviving triangle is re-written to the representatives, rotated so that the smallest index comes
//      first, and emitted unless an identical triangle has been emitted before.
//
// The output length is whatever survives — at most the target, data-dependent — and that length is what feeds
// index_len[lod], hence indexCount and the running firstIndex of cull_pass (cull_pipeline.rs:544-558).
// All arithmetic is binary32 in the source's operation order; build with -ffp-contract=off.
#include "simplify_sloppy.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <unordered_set>

namespace renderer {
namespace gltf {
namespace {

struct Vec3 { float x, y, z; };

struct Quadric {
  float a00, a11, a22;
  float a10, a20, a21;
  float b0, b1, b2, c;
  float w;
};

void rescale_positions(std::vector<Vec3>& result, const float* xyz, size_t vertex_count) {
  float minv[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  float maxv[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = 0; i < vertex_count; ++i) {
    const float* v = xyz + i * 3;
    result[i] = Vec3{v[0], v[1], v[2]};
    for (int j = 0; j < 3; ++j) {
      const float vj = v[j];
      minv[j] = minv[j] > vj ? vj : minv[j];
      maxv[j] = maxv[j] < vj ? vj : maxv[j];
    }
  }
  float extent = 0.f;
  extent = (maxv[0] - minv[0]) < extent ? extent : (maxv[0] - minv[0]);
  extent = (maxv[1] - minv[1]) < extent ? extent : (maxv[1] - minv[1]);
  extent = (maxv[2] - minv[2]) < extent ? extent : (maxv[2] - minv[2]);
  const float scale = extent == 0 ? 0.f : 1.f / extent;
  for (size_t i = 0; i < vertex_count; ++i) {
    result[i].x = (result[i].x - minv[0]) * scale;
    result[i].y = (result[i].y - minv[1]) * scale;
    result[i].z = (result[i].z - minv[2]) * scale;
  }
}

// float -> int as the compiled library does it (cvttss2si): out-of-range and NaN give INT_MIN instead of C++'s
// undefined behaviour (a non-finite position ends up in cell -2^31..., consistently)
int to_int(float v) {
  if (!(v > -2147483904.0f && v < 2147483648.0f)) return (-2147483647 - 1);
  return (int)v;
}

void compute_vertex_ids(std::vector<uint32_t>& ids, const std::vector<Vec3>& p, int grid_size) {
  const float cell_scale = (float)(grid_size - 1);
  for (size_t i = 0; i < p.size(); ++i) {
    const int xi = to_int(p[i].x * cell_scale + 0.5f);
    const int yi = to_int(p[i].y * cell_scale + 0.5f);
    const int zi = to_int(p[i].z * cell_scale + 0.5f);
    ids[i] = ((uint32_t)xi << 20) | ((uint32_t)yi << 10) | (uint32_t)zi;
  }
}

size_t count_triangles(const std::vector<uint32_t>& ids, const uint32_t* indices, size_t index_count) {
  size_t result = 0;
  for (size_t i = 0; i < index_count; i += 3) {
    const uint32_t id0 = ids[indices[i + 0]], id1 = ids[indices[i + 1]], id2 = ids[indices[i + 2]];
    result += (id0 != id1) & (id0 != id2) & (id1 != id2);
  }
  return result;
}

// three point interpolation from the "revenge of interpolation search" paper, as the library has it
float interpolate(float y, float x0, float y0, float x1, float y1, float x2, float y2) {
  const float num = (y1 - y) * (x1 - x2) * (x1 - x0) * (y2 - y0);
  const float den = (y2 - y) * (x1 - x2) * (y0 - y1) + (y0 - y) * (x1 - x0) * (y1 - y2);
  return x1 + num / den;
}

float normalize(Vec3& v) {
  const float length = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  if (length > 0) {
    v.x /= length;
    v.y /= length;
    v.z /= length;
  }
  return length;
}

void quadric_from_plane(Quadric& q, float a, float b, float c, float d, float w) {
  const float aw = a * w, bw = b * w, cw = c * w, dw = d * w;
  q.a00 = a * aw; q.a11 = b * bw; q.a22 = c * cw;
  q.a10 = a * bw; q.a20 = a * cw; q.a21 = b * cw;
  q.b0 = a * dw; q.b1 = b * dw; q.b2 = c * dw;
  q.c = d * dw;
  q.w = w;
}

void quadric_from_triangle(Quadric& q, const Vec3& p0, const Vec3& p1, const Vec3& p2, float weight) {
  const Vec3 p10 = {p1.x - p0.x, p1.y - p0.y, p1.z - p0.z};
  const Vec3 p20 = {p2.x - p0.x, p2.y - p0.y, p2.z - p0.z};
  Vec3 normal = {p10.y * p20.z - p10.z * p20.y, p10.z * p20.x - p10.x * p20.z, p10.x * p20.y - p10.y * p20.x};
  const float area = normalize(normal);
  const float distance = normal.x * p0.x + normal.y * p0.y + normal.z * p0.z;
  // sqrt(area) so that the error scales linearly
  quadric_from_plane(q, normal.x, normal.y, normal.z, -distance, std::sqrt(area) * weight);
}

void quadric_add(Quadric& q, const Quadric& r) {
  q.a00 += r.a00; q.a11 += r.a11; q.a22 += r.a22;
  q.a10 += r.a10; q.a20 += r.a20; q.a21 += r.a21;
  q.b0 += r.b0; q.b1 += r.b1; q.b2 += r.b2;
  q.c += r.c;
  q.w += r.w;
}

float quadric_error(const Quadric& q, const Vec3& v) {
  float rx = q.b0, ry = q.b1, rz = q.b2;
  rx += q.a10 * v.y; ry += q.a21 * v.z; rz += q.a20 * v.x;
  rx *= 2; ry *= 2; rz *= 2;
  rx += q.a00 * v.x; ry += q.a11 * v.y; rz += q.a22 * v.z;
  float r = q.c;
  r += rx * v.x; r += ry * v.y; r += rz * v.z;
  const float s = q.w == 0.f ? 0.f : 1.f / q.w;
  return std::fabs(r) * s;
}

struct TriangleKey {
  uint32_t a, b, c;
  bool operator==(const TriangleKey& o) const { return a == o.a && b == o.b && c == o.c; }
};
struct TriangleKeyHash {
  size_t operator()(const TriangleKey& t) const {
    uint64_t h = t.a * 0x9E3779B97F4A7C15ull;
    h ^= (t.b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    h ^= (t.c + 0x165667B1ull) * 0xD6E8FEB86659FD93ull;
    return (size_t)(h ^ (h >> 29));
  }
};

}  // namespace

std::vector<uint32_t> simplify_sloppy(const std::vector<uint32_t>& indices_in, const float* positions_xyz, size_t vertex_count,
                                      size_t target_index_count) {
  const uint32_t* indices = indices_in.data();
  const size_t index_count = indices_in.size() - indices_in.size() % 3;
  std::vector<uint32_t> destination;
  if (target_index_count > index_count) target_index_count = index_count;  // the library asserts this

  // we expect to get ~2 triangles/vertex in the output
  const size_t target_cell_count = target_index_count / 6;
  if (target_cell_count == 0) return destination;

  std::vector<Vec3> vertex_positions(vertex_count);
  rescale_positions(vertex_positions, positions_xyz, vertex_count);
  std::vector<uint32_t> vertex_ids(vertex_count);

  const int kInterpolationPasses = 5;
  // invariant: # of triangles in min_grid <= target_count
  int min_grid = 0, max_grid = 1025;
  size_t min_triangles = 0, max_triangles = index_count / 3;
  // instead of starting in the middle, guess: triangle count usually grows as a square of grid size
  int next_grid_size = to_int(std::sqrt((float)target_cell_count) + 0.5f);

  for (int pass = 0; pass < 10 + kInterpolationPasses; ++pass) {
    // the prediction is clamped so that the search converges
    int grid_size = next_grid_size;
    grid_size = (grid_size <= min_grid) ? min_grid + 1 : (grid_size >= max_grid) ? max_grid - 1 : grid_size;

    compute_vertex_ids(vertex_ids, vertex_positions, grid_size);
    const size_t triangles = count_triangles(vertex_ids, indices, index_count);

    const float tip = interpolate((float)(target_index_count / 3), (float)min_grid, (float)min_triangles, (float)grid_size,
                                  (float)triangles, (float)max_grid, (float)max_triangles);
    if (triangles <= target_index_count / 3) {
      min_grid = grid_size;
      min_triangles = triangles;
    } else {
      max_grid = grid_size;
      max_triangles = triangles;
    }
    if (triangles == target_index_count / 3 || max_grid - min_grid <= 1) break;

    // interpolation search first (usually converges faster), bisection after a few iterations (O(log N) worst case)
    next_grid_size = (pass < kInterpolationPasses) ? to_int(tip + 0.5f) : (min_grid + max_grid) / 2;
  }
  if (min_triangles == 0) return destination;

  // vertex -> cell: all vertices with the same quantised position share a cell; cells numbered by first appearance
  compute_vertex_ids(vertex_ids, vertex_positions, min_grid);
  std::vector<uint32_t> vertex_cells(vertex_count);
  size_t cell_count = 0;
  {
    std::unordered_map<uint32_t, uint32_t> cell_of_id;
    cell_of_id.reserve(vertex_count * 2);
    for (size_t i = 0; i < vertex_count; ++i) {
      auto it = cell_of_id.find(vertex_ids[i]);
      if (it == cell_of_id.end()) {
        cell_of_id.emplace(vertex_ids[i], (uint32_t)cell_count);
        vertex_cells[i] = (uint32_t)cell_count++;
      } else {
        vertex_cells[i] = it->second;
      }
    }
  }

  // a quadric for each target cell
  std::vector<Quadric> cell_quadrics(cell_count);
  std::memset(cell_quadrics.data(), 0, cell_count * sizeof(Quadric));
  for (size_t i = 0; i < index_count; i += 3) {
    const uint32_t i0 = indices[i + 0], i1 = indices[i + 1], i2 = indices[i + 2];
    const uint32_t c0 = vertex_cells[i0], c1 = vertex_cells[i1], c2 = vertex_cells[i2];
    const bool single_cell = (c0 == c1) & (c0 == c2);
    Quadric q;
    quadric_from_triangle(q, vertex_positions[i0], vertex_positions[i1], vertex_positions[i2], single_cell ? 3.f : 1.f);
    if (single_cell) {
      quadric_add(cell_quadrics[c0], q);
    } else {
      quadric_add(cell_quadrics[c0], q);
      quadric_add(cell_quadrics[c1], q);
      quadric_add(cell_quadrics[c2], q);
    }
  }

  // for each target cell, the vertex with the minimal error
  std::vector<uint32_t> cell_remap(cell_count, 0xffffffffu);
  std::vector<float> cell_errors(cell_count, 0.f);
  for (size_t i = 0; i < vertex_count; ++i) {
    const uint32_t cell = vertex_cells[i];
    const float error = quadric_error(cell_quadrics[cell], vertex_positions[i]);
    if (cell_remap[cell] == 0xffffffffu || cell_errors[cell] > error) {
      cell_remap[cell] = (uint32_t)i;
      cell_errors[cell] = error;
    }
  }

  // collapse triangles; cells very frequently generate the same triangle more than once: emit each once
  std::unordered_set<TriangleKey, TriangleKeyHash> seen;
  seen.reserve(min_triangles * 2);
  for (size_t i = 0; i < index_count; i += 3) {
    const uint32_t c0 = vertex_cells[indices[i + 0]], c1 = vertex_cells[indices[i + 1]], c2 = vertex_cells[indices[i + 2]];
    if (c0 != c1 && c0 != c2 && c1 != c2) {
      uint32_t a = cell_remap[c0], b = cell_remap[c1], c = cell_remap[c2];
      if (b < a && b < c) {
        const uint32_t t = a;
        a = b, b = c, c = t;
      } else if (c < a && c < b) {
        const uint32_t t = c;
        c = b, b = a, a = t;
      }
      if (seen.insert(TriangleKey{a, b, c}).second) {
        destination.push_back(a);
        destination.push_back(b);
        destination.push_back(c);
      }
    }
  }
  return destination;
}

}  // namespace gltf
}  // namespace renderer
