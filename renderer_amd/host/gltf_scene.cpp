// gltf_scene.cpp — see gltf_scene.hpp.
#include "gltf_scene.hpp"
#include "simplify_sloppy.hpp"

#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace renderer {
namespace gltf {
namespace {

// ------------------------------------------------------------------ minimal JSON
struct Json {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;

  const Json* get(const char* key) const {
    if (kind != Object) return nullptr;
    for (const auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  bool has(const char* key) const { return get(key) != nullptr; }
  double number_or(const char* key, double d) const {
    const Json* j = get(key);
    return (j && j->kind == Number) ? j->num : d;
  }
  size_t size() const { return kind == Array ? arr.size() : 0; }
};

// A JSON number used as an index, count, offset or stride: finite, non-negative, integral, < 2^40.
// (Converting a negative, NaN or huge double to size_t is undefined behaviour, and every such value
// comes straight from an untrusted file.)
size_t to_index(const Json& j, const char* what) {
  if (j.kind != Json::Number || !(j.num >= 0.0) || j.num >= 1099511627776.0 || j.num != std::floor(j.num))
    throw std::runtime_error(std::string("gltf: ") + what + " is not a valid non-negative integer");
  return (size_t)j.num;
}
size_t index_or(const Json& obj, const char* key, size_t fallback) {
  const Json* j = obj.get(key);
  return j ? to_index(*j, key) : fallback;
}

struct Parser {
  const std::string& s;
  size_t i = 0;
  int depth = 0;  // arrays + objects open at this point: the parser recurses, so untrusted nesting is capped
  static constexpr int kMaxDepth = 64;
  explicit Parser(const std::string& text) : s(text) {}
  struct Nest {
    Parser& p;
    explicit Nest(Parser& parser) : p(parser) {
      if (++p.depth > kMaxDepth) p.fail("nesting deeper than 64");
    }
    ~Nest() { --p.depth; }
  };
  [[noreturn]] void fail(const char* what) const {
    throw std::runtime_error(std::string("gltf json: ") + what + " at byte " + std::to_string(i));
  }
  void ws() {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\r' || s[i] == '\t')) ++i;
  }
  Json value() {
    ws();
    if (i >= s.size()) fail("unexpected end");
    const char c = s[i];
    if (c == '{') return object();
    if (c == '[') return array();
    if (c == '"') {
      Json j;
      j.kind = Json::String;
      j.str = string();
      return j;
    }
    if (!s.compare(i, 4, "true")) { i += 4; Json j; j.kind = Json::Bool; j.b = true; return j; }
    if (!s.compare(i, 5, "false")) { i += 5; Json j; j.kind = Json::Bool; return j; }
    if (!s.compare(i, 4, "null")) { i += 4; return Json(); }
    return number();
  }
  Json number() {
    const size_t start = i;
    while (i < s.size() && (std::isdigit((unsigned char)s[i]) || s[i] == '-' || s[i] == '+' || s[i] == '.' || s[i] == 'e' || s[i] == 'E')) ++i;
    if (start == i) fail("bad token");
    Json j;
    j.kind = Json::Number;
    j.num = std::strtod(s.substr(start, i - start).c_str(), nullptr);
    return j;
  }
  std::string string() {
    ++i;  // opening quote
    std::string out;
    while (i < s.size() && s[i] != '"') {
      if (s[i] == '\\') {
        ++i;
        if (i >= s.size()) fail("bad escape");
        switch (s[i]) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {  // keep BMP code points as UTF-8; enough for names and URIs
            if (i + 4 >= s.size()) fail("bad \\u");
            const unsigned cp = (unsigned)std::strtoul(s.substr(i + 1, 4).c_str(), nullptr, 16);
            i += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += s[i];
        }
        ++i;
      } else {
        out += s[i++];
      }
    }
    if (i >= s.size()) fail("unterminated string");
    ++i;
    return out;
  }
  Json array() {
    const Nest nest(*this);
    Json j;
    j.kind = Json::Array;
    ++i;
    ws();
    if (i < s.size() && s[i] == ']') { ++i; return j; }
    for (;;) {
      j.arr.push_back(value());
      ws();
      if (i >= s.size()) fail("unterminated array");
      if (s[i] == ',') { ++i; continue; }
      if (s[i] == ']') { ++i; return j; }
      fail("expected , or ]");
    }
  }
  Json object() {
    const Nest nest(*this);
    Json j;
    j.kind = Json::Object;
    ++i;
    ws();
    if (i < s.size() && s[i] == '}') { ++i; return j; }
    for (;;) {
      ws();
      if (i >= s.size() || s[i] != '"') fail("expected key");
      std::string key = string();
      ws();
      if (i >= s.size() || s[i] != ':') fail("expected :");
      ++i;
      j.obj.emplace_back(std::move(key), value());
      ws();
      if (i >= s.size()) fail("unterminated object");
      if (s[i] == ',') { ++i; continue; }
      if (s[i] == '}') { ++i; return j; }
      fail("expected , or }");
    }
  }
};

// ------------------------------------------------------------------ buffers
std::string read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("gltf: cannot open " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

std::string base64_decode(const std::string& in, size_t start) {
  static int table[256];
  static bool init = false;
  if (!init) {
    for (int k = 0; k < 256; ++k) table[k] = -1;
    const char* abc = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    for (int k = 0; k < 64; ++k) table[(unsigned char)abc[k]] = k;
    init = true;
  }
  std::string out;
  unsigned acc = 0;
  int bits = 0;
  for (size_t k = start; k < in.size(); ++k) {
    const int v = table[(unsigned char)in[k]];
    if (v < 0) continue;  // '=', whitespace
    acc = (acc << 6) | (unsigned)v;
    bits += 6;
    if (bits >= 8) {
      bits -= 8;
      out += (char)((acc >> bits) & 0xFF);
    }
  }
  return out;
}

std::string dirname_of(const std::string& path) {
  const size_t slash = path.find_last_of('/');
  return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
}

struct Document {
  Json root;
  std::vector<std::string> buffers;
};

Document open_document(const std::string& path) {
  Document d;
  std::string text = read_file(path);
  std::string glb_bin;
  bool have_glb_bin = false;
  if (text.size() >= 20 && !text.compare(0, 4, "glTF")) {  // .glb: 12-byte header, JSON chunk, optional BIN chunk
    uint32_t json_len;
    std::memcpy(&json_len, &text[12], 4);
    if (text.compare(16, 4, "JSON") || 20 + (size_t)json_len > text.size()) throw std::runtime_error("gltf: malformed glb");
    const std::string json = text.substr(20, json_len);
    size_t at = 20 + json_len;
    if (at + 8 <= text.size()) {
      uint32_t bin_len;
      std::memcpy(&bin_len, &text[at], 4);
      if (!text.compare(at + 4, 3, "BIN") && at + 8 + bin_len <= text.size()) {
        glb_bin = text.substr(at + 8, bin_len);
        have_glb_bin = true;
      }
    }
    text = json;
  }
  Parser p(text);
  d.root = p.value();
  if (d.root.kind != Json::Object) throw std::runtime_error("gltf: top level is not an object");
  const Json* buffers = d.root.get("buffers");
  for (size_t k = 0; buffers && k < buffers->size(); ++k) {
    const Json& b = buffers->arr[k];
    const Json* uri = b.get("uri");
    if (!uri) {
      if (k == 0 && have_glb_bin) d.buffers.push_back(glb_bin);
      else throw std::runtime_error("gltf: buffer without uri");
    } else if (uri->kind != Json::String || uri->str.empty() || uri->str.find("://") != std::string::npos) {
      throw std::runtime_error("gltf: buffer uri must be a data: uri or a relative file path");
    } else if (!uri->str.compare(0, 5, "data:")) {
      const size_t comma = uri->str.find(',');
      if (comma == std::string::npos) throw std::runtime_error("gltf: bad data uri");
      d.buffers.push_back(base64_decode(uri->str, comma + 1));
    } else {
      // an untrusted .gltf must not make the extractor read arbitrary local files into its output: the buffer has to
      // live in (or below) the asset's own directory — no absolute path, no scheme, no ".." component, and the
      // resolved path (symbolic links followed) must still be inside that directory
      const std::string& u = uri->str;
      bool bad = u[0] == '/' || u[0] == '\\' || u.find(':') != std::string::npos;
      for (size_t at = 0; !bad && at <= u.size();) {
        size_t end = u.find_first_of("/\\", at);
        if (end == std::string::npos) end = u.size();
        if (u.compare(at, end - at, "..") == 0) bad = true;
        at = end + 1;
      }
      if (bad) throw std::runtime_error("gltf: buffer uri must be a data: uri or a file path relative to the asset, inside its directory");
      const std::string dir = dirname_of(path), full = dir + "/" + u;
      char real_dir[PATH_MAX], real_file[PATH_MAX];
      if (!realpath(dir.c_str(), real_dir) || !realpath(full.c_str(), real_file)) throw std::runtime_error("gltf: cannot open " + full);
      const size_t dl = std::strlen(real_dir);
      if (std::strncmp(real_dir, real_file, dl) != 0 || (real_file[dl] != '/' && dl > 1))
        throw std::runtime_error("gltf: buffer uri resolves outside the asset's directory");
      d.buffers.push_back(read_file(real_file));
    }
  }
  return d;
}

// ------------------------------------------------------------------ accessors
struct AccessorView {
  const unsigned char* data = nullptr;
  size_t count = 0, stride = 0;
  int component_type = 0, components = 0;
};

int component_size(int type) {
  switch (type) {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
  }
  throw std::runtime_error("gltf: unknown componentType");
}

AccessorView view_of(const Document& d, size_t accessor_index) {
  const Json* accessors = d.root.get("accessors");
  if (!accessors || accessor_index >= accessors->size()) throw std::runtime_error("gltf: accessor index out of range");
  const Json& a = accessors->arr[accessor_index];
  if (a.has("sparse")) throw std::runtime_error("gltf: sparse accessors are not supported");
  const Json* bv_index = a.get("bufferView");
  if (!bv_index) throw std::runtime_error("gltf: accessor without bufferView");
  const Json* views = d.root.get("bufferViews");
  if (!views || to_index(*bv_index, "bufferView") >= views->size()) throw std::runtime_error("gltf: bufferView index out of range");
  const Json& bv = views->arr[to_index(*bv_index, "bufferView")];
  const size_t buffer = index_or(bv, "buffer", 0);
  if (buffer >= d.buffers.size()) throw std::runtime_error("gltf: buffer index out of range");
  AccessorView v;
  v.component_type = (int)index_or(a, "componentType", 0);
  const Json* type = a.get("type");
  const std::string t = type ? type->str : "";
  v.components = t == "SCALAR" ? 1 : t == "VEC2" ? 2 : t == "VEC3" ? 3 : t == "VEC4" ? 4 : 0;
  if (!v.components) throw std::runtime_error("gltf: unsupported accessor type " + t);
  v.count = index_or(a, "count", 0);
  const size_t elem = (size_t)component_size(v.component_type) * v.components;
  v.stride = index_or(bv, "byteStride", 0);
  if (!v.stride) v.stride = elem;
  const size_t offset = index_or(bv, "byteOffset", 0) + index_or(a, "byteOffset", 0);  // each < 2^40: no wrap
  const std::string& buf = d.buffers[buffer];
  // count, stride < 2^40 would wrap a 64-bit product: bound the count by what the buffer can hold first
  if (v.count && (offset > buf.size() || elem > buf.size() - offset || (v.count - 1) > (buf.size() - offset - elem) / v.stride))
    throw std::runtime_error("gltf: accessor overruns its buffer");
  v.data = (const unsigned char*)buf.data() + offset;
  return v;
}

// ------------------------------------------------------------------ node transform
// gltf crate Transform::decomposed(): TRS as stored; a matrix is split into translation,
// per-axis scale (column lengths, z signed by the determinant) and a quaternion.
void decompose(const Json& node, float t[3], float r[4], float s[3]) {
  t[0] = t[1] = t[2] = 0;
  r[0] = r[1] = r[2] = 0; r[3] = 1;
  s[0] = s[1] = s[2] = 1;
  if (const Json* m = node.get("matrix")) {
    if (m->size() != 16) throw std::runtime_error("gltf: node.matrix needs 16 numbers");
    double c[3][3];
    for (int col = 0; col < 3; ++col)
      for (int row = 0; row < 3; ++row) c[col][row] = m->arr[col * 4 + row].num;
    for (int k = 0; k < 3; ++k) t[k] = (float)m->arr[12 + k].num;
    const double det = c[0][0] * (c[1][1] * c[2][2] - c[2][1] * c[1][2]) - c[1][0] * (c[0][1] * c[2][2] - c[2][1] * c[0][2]) +
                       c[2][0] * (c[0][1] * c[1][2] - c[1][1] * c[0][2]);
    double len[3];
    for (int k = 0; k < 3; ++k) len[k] = std::sqrt(c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2]);
    if (det < 0) len[2] = -len[2];
    for (int k = 0; k < 3; ++k) {
      s[k] = (float)len[k];
      for (int row = 0; row < 3; ++row) c[k][row] = len[k] != 0 ? c[k][row] / len[k] : 0.0;
    }
    // rotation matrix (columns c[k]) -> quaternion, trace method
    const double m00 = c[0][0], m11 = c[1][1], m22 = c[2][2];
    const double trace = m00 + m11 + m22;
    double x, y, z, w;
    if (trace > 0) {
      const double q = std::sqrt(trace + 1.0) * 2;
      w = q / 4; x = (c[1][2] - c[2][1]) / q; y = (c[2][0] - c[0][2]) / q; z = (c[0][1] - c[1][0]) / q;
    } else if (m00 > m11 && m00 > m22) {
      const double q = std::sqrt(1.0 + m00 - m11 - m22) * 2;
      w = (c[1][2] - c[2][1]) / q; x = q / 4; y = (c[1][0] + c[0][1]) / q; z = (c[2][0] + c[0][2]) / q;
    } else if (m11 > m22) {
      const double q = std::sqrt(1.0 + m11 - m00 - m22) * 2;
      w = (c[2][0] - c[0][2]) / q; x = (c[1][0] + c[0][1]) / q; y = q / 4; z = (c[2][1] + c[1][2]) / q;
    } else {
      const double q = std::sqrt(1.0 + m22 - m00 - m11) * 2;
      w = (c[0][1] - c[1][0]) / q; x = (c[2][0] + c[0][2]) / q; y = (c[2][1] + c[1][2]) / q; z = q / 4;
    }
    r[0] = (float)x; r[1] = (float)y; r[2] = (float)z; r[3] = (float)w;
    return;
  }
  if (const Json* j = node.get("translation"))
    for (int k = 0; k < 3 && (size_t)k < j->size(); ++k) t[k] = (float)j->arr[k].num;
  if (const Json* j = node.get("rotation"))
    for (int k = 0; k < 4 && (size_t)k < j->size(); ++k) r[k] = (float)j->arr[k].num;  // glTF [x,y,z,w] = [i,j,k,w]
  if (const Json* j = node.get("scale"))
    for (int k = 0; k < 3 && (size_t)k < j->size(); ++k) s[k] = (float)j->arr[k].num;
}

bool has_base_color_texture(const Document& d, const Json& primitive) {
  const Json* mat_index = primitive.get("material");
  const Json* materials = d.root.get("materials");
  if (!mat_index || !materials || to_index(*mat_index, "material") >= materials->size()) return false;
  const Json* pbr = materials->arr[to_index(*mat_index, "material")].get("pbrMetallicRoughness");
  return pbr && pbr->has("baseColorTexture");
}

void visit_node(const Document& d, size_t node_index, Scene& out, int depth, size_t& visits) {
  const Json* nodes = d.root.get("nodes");
  if (!nodes || node_index >= nodes->size()) throw std::runtime_error("gltf: node index out of range");
  if (depth > 256) throw std::runtime_error("gltf: node hierarchy too deep (cycle?)");
  // a node graph that is not a tree (shared or cyclic children) can fan out exponentially below the depth limit
  if (++visits > (size_t)1 << 20) throw std::runtime_error("gltf: more than 2^20 node visits (children shared or cyclic?)");
  const Json& node = nodes->arr[node_index];
  if (const Json* mesh_index = node.get("mesh")) {
    const Json* meshes = d.root.get("meshes");
    if (!meshes || to_index(*mesh_index, "mesh") >= meshes->size()) throw std::runtime_error("gltf: mesh index out of range");
    const Json& mesh = meshes->arr[to_index(*mesh_index, "mesh")];
    const Json* primitives = mesh.get("primitives");
    for (size_t p = 0; primitives && p < primitives->size(); ++p) {
      const Json& prim = primitives->arr[p];
      ++out.primitives_seen;
      if (!has_base_color_texture(d, prim)) {  // scene_loader.rs:659-667
        ++out.skipped_no_base_color;
        continue;
      }
      const Json* attributes = prim.get("attributes");
      const Json* pos_acc = attributes ? attributes->get("POSITION") : nullptr;
      if (!pos_acc) throw std::runtime_error("gltf: primitive without POSITION");  // "failed to load positions"
      const AccessorView positions = view_of(d, to_index(*pos_acc, "POSITION"));
      if (positions.component_type != 5126 || positions.components != 3) throw std::runtime_error("gltf: POSITION must be float VEC3");
      if (positions.count < 100) {  // :677-679
        ++out.skipped_small;
        continue;
      }
      const Json& pos_json = d.root.get("accessors")->arr[to_index(*pos_acc, "POSITION")];
      const Json* mn = pos_json.get("min");
      const Json* mx = pos_json.get("max");
      if (!mn || !mx || mn->size() != 3 || mx->size() != 3) throw std::runtime_error("gltf: POSITION accessor needs min/max");
      const Json* idx_acc = prim.get("indices");
      if (!idx_acc) throw std::runtime_error("gltf: primitive without indices");  // "failed to load indices"
      const AccessorView idx = view_of(d, to_index(*idx_acc, "indices"));
      if (idx.components != 1) throw std::runtime_error("gltf: indices must be SCALAR");
      if (positions.count > 0x7fffffffull || out.vertices.size() / 3 + positions.count > 0x7fffffffull || idx.count > 0xffffffffull)
        throw std::runtime_error("gltf: geometry too large for 32-bit offsets");

      MipMesh m{};
      for (int k = 0; k < 3; ++k) {
        m.aabb_min[k] = (float)mn->arr[k].num;
        m.aabb_max[k] = (float)mx->arr[k].num;
      }
      m.vertex_offset = (int32_t)(out.vertices.size() / 3);
      for (size_t v = 0; v < positions.count; ++v) {
        float xyz[3];
        std::memcpy(xyz, positions.data + v * positions.stride, 12);
        out.vertices.insert(out.vertices.end(), xyz, xyz + 3);
      }
      std::vector<uint32_t> lod0(idx.count);
      for (size_t k = 0; k < idx.count; ++k) {
        const unsigned char* e = idx.data + k * idx.stride;
        uint32_t v = 0;
        if (idx.component_type == 5121) v = e[0];
        else if (idx.component_type == 5123) { uint16_t h; std::memcpy(&h, e, 2); v = h; }
        else if (idx.component_type == 5125) std::memcpy(&v, e, 4);
        else throw std::runtime_error("gltf: indices must be unsigned");
        // the per-triangle stage gathers vertices[vertex_offset + index] unchecked on the device
        if (v >= positions.count) throw std::runtime_error("gltf: index " + std::to_string(v) + " outside the primitive's " + std::to_string(positions.count) + " positions");
        lod0[k] = v;
      }
      // LOD chain (:739-753): LOD 0 + up to five levels from simplify_sloppy, each kept only if it came out shorter
      // than LOD 0 and non-empty. The length of a level is whatever the simplifier returns (at most the target), so
      // index_len[lod] — hence indexCount and the running firstIndex of cull_pass — is data-dependent.
      std::vector<std::vector<uint32_t>> lods;
      lods.push_back(lod0);
      const float* prim_positions = out.vertices.data() + (size_t)m.vertex_offset * 3;
      for (int x = 1; x < 6; ++x) {
        const float factor = std::pow(0.5f, (float)x);
        const size_t target = (size_t)((float)lod0.size() * factor);  // `(indices.len() as f32 * factor) as usize`
        std::vector<uint32_t> res = simplify_sloppy(lod0, prim_positions, (size_t)positions.count, target);
        if (res.size() < lod0.size() && !res.empty()) lods.push_back(std::move(res));
      }
      m.n_lods = (uint32_t)lods.size();
      for (size_t l = 0; l < lods.size(); ++l) {
        m.index_len[l] = (uint32_t)lods[l].size();
        m.index_offset[l] = (uint32_t)out.indices.size();
        out.indices.insert(out.indices.end(), lods[l].begin(), lods[l].end());
      }
      float t[3], r[4], s[3];
      decompose(node, t, r, s);
      out.pos_xyz.insert(out.pos_xyz.end(), t, t + 3);
      out.rot_ijkw.insert(out.rot_ijkw.end(), r, r + 4);
      out.scale.push_back(s[0]);  // scale: scale[0], :765
      out.mesh_id.push_back((uint32_t)out.meshes.size());
      out.meshes.push_back(m);
      const Json* name = node.get("name");
      out.entity_names.push_back(name ? name->str : std::string());
    }
  }
  if (const Json* children = node.get("children"))
    for (size_t k = 0; k < children->size(); ++k) visit_node(d, to_index(children->arr[k], "child"), out, depth + 1, visits);
}

}  // namespace

Scene load(const std::string& path) {
  const Document d = open_document(path);
  Scene out;
  size_t visits = 0;
  const Json* scenes = d.root.get("scenes");
  for (size_t s = 0; scenes && s < scenes->size(); ++s) {  // every scene, every root node (:137-142)
    const Json* roots = scenes->arr[s].get("nodes");
    for (size_t k = 0; roots && k < roots->size(); ++k) visit_node(d, to_index(roots->arr[k], "root node"), out, 0, visits);
  }
  return out;
}

}  // namespace gltf
}  // namespace renderer
