// scene_file.cpp — the loader3d front-end: `.scene` grammar (examples/loader3d.rs:214-906), MTL
// (src/mtl.rs), OBJ (src/obj.rs), texture decode (src/texture2d.rs:99-177), camera set-up
// (examples/loader3d.rs:68-79).  Produces the POD descriptors of include/nrays_abi.h.
#include "host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

namespace nrays_host {
namespace {

std::vector<std::string> words(const std::string& line) { // split_whitespace / obj::split_words
    std::vector<std::string> w; std::istringstream is(line); std::string t;
    while (is >> t) w.push_back(t);
    return w;
}
std::string join(const std::vector<std::string>& w, size_t from) { // parse_name: rest of line joined by ' '
    std::string r;
    for (size_t i = from; i < w.size(); ++i) { if (i > from) r += ' '; r += w[i]; }
    return r;
}
[[noreturn]] void fail(size_t line, const std::string& msg) { throw std::runtime_error("At line " + std::to_string(line) + ": " + msg); }
double num(size_t l, const std::vector<std::string>& w, size_t i, size_t expected) {
    if (i >= w.size()) fail(l, std::to_string(expected) + " components were expected, found " + std::to_string(w.size() - 1) + ".");
    try { size_t p = 0; double v = std::stod(w[i], &p); if (p != w[i].size()) throw 0; return v; }
    catch (...) { fail(l, "failed to parse `" + w[i] + "' as a f64."); }
}
float numf(size_t l, const std::vector<std::string>& w, size_t i, size_t expected) {
    if (i >= w.size()) fail(l, std::to_string(expected) + " components were expected, found " + std::to_string(w.size() - 1) + ".");
    try { size_t p = 0; float v = std::stof(w[i], &p); if (p != w[i].size()) throw 0; return v; }
    catch (...) { fail(l, "failed to parse `" + w[i] + "' as a f32."); }
}
std::string read_file(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Unable to find the file: " + path);
    std::stringstream ss; ss << f.rdbuf(); return ss.str();
}
std::string join_path(const std::string& a, const std::string& b) {
    if (a.empty() || (!b.empty() && b[0] == '/')) return b;
    return a.back() == '/' ? a + b : a + "/" + b;
}
std::string dir_of(const std::string& p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? std::string(".") : p.substr(0, k); }
bool file_exists(const std::string& p) { std::ifstream f(p); return (bool)f; }

// Procedural stand-in for scenes/media/globe.png (absent from the reference tree);
// identical to tests/scenes_util.py::globe_texture (1024x512, row 0 = bottom).
std::shared_ptr<ImageData> globe_standin() {
    auto d = std::make_shared<ImageData>();
    const uint32_t w = 1024, h = 512;
    d->width = w; d->height = h; d->format = NRAYS_TEXEL_RGBA8; d->bytes.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) {
        uint32_t chk = ((x * 24 / w) + (y * 12 / h)) % 2;
        uint8_t* p = &d->bytes[((size_t)y * w + x) * 4];
        p[0] = (uint8_t)(40 + 180 * chk); p[1] = (uint8_t)(x * 255 / (w - 1)); p[2] = (uint8_t)(y * 255 / (h - 1)); p[3] = 255;
    }
    return d;
}

} // namespace

// ---------------------------------------------------------------------------------- textures --
std::shared_ptr<ImageData> decode_texture(const Image8& img, bool opacity) {
    auto d = std::make_shared<ImageData>();
    d->width = img.width; d->height = img.height;
    const int c = img.channels;
    size_t n = (size_t)img.width * img.height;
    auto src = [&](uint32_t x, uint32_t y, int k) { return img.data[(((size_t)(img.height - 1 - y)) * img.width + x) * c + k]; }; // Y flip, texture2d.rs:99-107
    if (c == 2) { // r*g products are not u8-representable: keep the reference's f32 texels
        d->format = NRAYS_TEXEL_RGBA32F; d->bytes.resize(n * 16);
        float* o = (float*)d->bytes.data();
        for (uint32_t y = 0; y < img.height; ++y) for (uint32_t x = 0; x < img.width; ++x, o += 4) {
            float r = (float)src(x, y, 0) / 255.0f, g = (float)src(x, y, 1) / 255.0f;
            if (opacity) { o[0] = o[1] = o[2] = 1.0f; o[3] = g * r; } else { o[0] = o[1] = o[2] = r * g; o[3] = 1.0f; }
        }
        return d;
    }
    if (c != 1 && c != 3 && c != 4) throw std::runtime_error("Image depth " + std::to_string(c) + " not suported.");
    d->format = NRAYS_TEXEL_RGBA8; d->bytes.resize(n * 4);
    uint8_t* o = d->bytes.data();
    for (uint32_t y = 0; y < img.height; ++y) for (uint32_t x = 0; x < img.width; ++x, o += 4) {
        if (opacity) { o[0] = o[1] = o[2] = 255; o[3] = c == 4 ? src(x, y, 3) : src(x, y, 0); } // depth 3 opacity uses r (texture2d.rs:146-148)
        else if (c == 1) { o[0] = o[1] = o[2] = src(x, y, 0); o[3] = 255; }
        else { o[0] = src(x, y, 0); o[1] = src(x, y, 1); o[2] = src(x, y, 2); o[3] = 255; }
    }
    return d;
}

// --------------------------------------------------------------------------------------- MTL --
std::vector<MtlMaterial> parse_mtl_file(const std::string& path) {
    std::string text = read_file(path);
    std::vector<MtlMaterial> res; MtlMaterial cur;
    std::istringstream is(text); std::string line; size_t l = 0;
    for (; std::getline(is, line); ++l) {
        auto w = words(line);
        if (w.empty() || w[0][0] == '#' || w.size() < 2) continue; // mtl.rs:37-47
        const std::string& t = w[0];
        if (t == "newmtl") { if (!cur.name.empty()) res.push_back(cur); cur = MtlMaterial(); cur.name = join(w, 1); }
        else if (t == "Ka") for (int k = 0; k < 3; ++k) cur.ambiant[k] = numf(l, w, 1 + k, 3);
        else if (t == "Kd") for (int k = 0; k < 3; ++k) cur.diffuse[k] = numf(l, w, 1 + k, 3);
        else if (t == "Ks") for (int k = 0; k < 3; ++k) cur.specular[k] = numf(l, w, 1 + k, 3);
        else if (t == "Ns") cur.shininess = numf(l, w, 1, 1);
        else if (t == "d") cur.alpha = numf(l, w, 1, 1);
        else if (t == "map_Ka") cur.ambiant_texture = join(w, 1);
        else if (t == "map_Kd") cur.diffuse_texture = join(w, 1);
        else if (t == "map_Ks") cur.specular_texture = join(w, 1);
        else if (t == "map_d" || t == "map_opacity") cur.opacity_map = join(w, 1);
    }
    if (!cur.name.empty()) res.push_back(cur);
    return res;
}

// --------------------------------------------------------------------------------------- OBJ --
ObjFile parse_obj_file(const std::string& path, const std::string& mtl_base_dir) {
    std::string text = read_file(path);
    struct Id { size_t x, y, z; bool operator<(const Id& o) const { return x != o.x ? x < o.x : (y != o.y ? y < o.y : z < o.z); } };
    const size_t kMax = (size_t)std::numeric_limits<int32_t>::max(); // Bounded::max_value() as usize
    std::vector<float> coords, uvs;
    size_t nnormals = 0;
    std::map<std::string, size_t> groups; std::vector<std::string> group_names;
    std::vector<std::vector<Id>> groups_ids(1);
    std::map<size_t, MtlMaterial> group2mtl;
    std::map<std::string, MtlMaterial> mtllib;
    bool ignore_uvs = false, ignore_normals = false, have_mtl = false;
    MtlMaterial curr_mtl;
    size_t curr_group = 0;
    groups[""] = 0; group_names.push_back("");
    auto parse_g = [&](const std::string& suffix, const std::string& prefix) { // obj.rs:312-332
        std::string name = suffix.empty() ? prefix : prefix + "/" + suffix;
        auto it = groups.find(name);
        if (it != groups.end()) return it->second;
        groups_ids.emplace_back(); group_names.push_back(name);
        return groups[name] = groups_ids.size() - 1;
    };
    std::istringstream is(text); std::string line; size_t l = 0;
    for (; std::getline(is, line); ++l) {
        auto w = words(line);
        if (w.empty() || w[0][0] == '#') continue;
        const std::string& t = w[0];
        if (t == "v") { for (int k = 0; k < 3; ++k) coords.push_back(numf(l, w, 1 + k, 3)); }
        else if (t == "vn") { if (!ignore_normals) { for (int k = 0; k < 3; ++k) numf(l, w, 1 + k, 3); ++nnormals; } }
        else if (t == "vt") { if (!ignore_uvs) { uvs.push_back(numf(l, w, 1, 2)); uvs.push_back(numf(l, w, 2, 2)); } }
        else if (t == "f") { // parse_f, obj.rs:207-291 (including its on-the-fly "fan" indexing)
            std::vector<Id>& g = groups_ids[curr_group];
            size_t i = 0;
            for (size_t wi = 1; wi < w.size(); ++wi) {
                long ids[3] = {(long)kMax, (long)kMax, (long)kMax};
                std::string word = w[wi]; size_t start = 0; int comp = 0;
                while (comp < 3) {
                    size_t slash = word.find('/', start);
                    std::string part = word.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
                    if (comp == 0 || !part.empty()) {
                        try { size_t p = 0; long v = std::stol(part, &p); if (p != part.size()) throw 0; ids[comp] = v - 1; }
                        catch (...) { fail(l, "failed to parse `" + part + "' as a i32."); }
                    }
                    if (slash == std::string::npos) break;
                    start = slash + 1; ++comp;
                }
                if (i > 2) { Id p1 = g[g.size() - i], p2 = g[g.size() - 1]; g.push_back(p1); g.push_back(p2); }
                if (ids[1] == (long)kMax) ignore_uvs = true;
                if (ids[2] == (long)kMax) ignore_normals = true;
                Id id;
                id.x = ids[0] < 0 ? (size_t)((long)(coords.size() / 3) + ids[0] + 1) : (size_t)ids[0];
                id.y = ids[1] < 0 ? (size_t)((long)(uvs.size() / 2) + ids[1] + 1) : (size_t)ids[1];
                id.z = ids[2] < 0 ? (size_t)((long)nnormals + ids[2] + 1) : (size_t)ids[2];
                g.push_back(id);
                ++i;
            }
            if (i < 2 && !g.empty()) for (size_t k = 0; k < 3 - i; ++k) g.push_back(g.back());
        }
        else if (t == "g") { curr_group = parse_g(join(w, 1), ""); if (have_mtl) group2mtl[curr_group] = curr_mtl; }
        else if (t == "mtllib") {
            std::string p = join_path(mtl_base_dir, join(w, 1));
            try { for (auto& m : parse_mtl_file(p)) mtllib[m.name] = m; } catch (const std::exception&) { /* warn, obj.rs:189 */ }
        }
        else if (t == "usemtl") { // parse_usemtl, obj.rs:122-168
            std::string mname = join(w, 1);
            if (mname != "None") {
                auto it = mtllib.find(mname);
                if (it == mtllib.end()) have_mtl = false;
                else if (!group2mtl.count(curr_group)) { group2mtl[curr_group] = it->second; curr_mtl = it->second; have_mtl = true; }
                else {
                    auto sw = words(std::to_string(curr_group) + mname);
                    size_t ng = parse_g(join(sw, 0), "auto_generated_group_");
                    group2mtl[ng] = it->second; curr_mtl = it->second; have_mtl = true; curr_group = ng;
                }
            } else have_mtl = false;
        }
    }
    // reformat, obj.rs:334-397; groups in order of first appearance (the reference iterates a HashMap: D-11)
    ObjFile out;
    std::map<Id, uint32_t> vt2id;
    bool use_uvs = !ignore_uvs && !uvs.empty();
    for (size_t gi = 0; gi < groups_ids.size(); ++gi) {
        const std::vector<Id>& g = groups_ids[gi];
        if (g.size() % 3 != 0) throw std::runtime_error("obj: face list of group `" + group_names[gi] + "' is not a multiple of 3");
        ObjGroup og; og.name = group_names[gi];
        auto mt = group2mtl.find(gi);
        if (mt != group2mtl.end()) { og.has_mtl = true; og.mtl = mt->second; }
        for (const Id& id : g) {
            Id key = id; if (ignore_uvs) key.y = kMax; if (ignore_normals) key.z = kMax;
            auto it = vt2id.find(key);
            uint32_t idx;
            if (it == vt2id.end()) {
                if (3 * id.x + 2 >= coords.size()) throw std::runtime_error("obj: vertex index out of range");
                idx = (uint32_t)(out.coords.size() / 3);
                for (int k = 0; k < 3; ++k) out.coords.push_back(coords[3 * id.x + k]);
                if (use_uvs) { if (2 * id.y + 1 >= uvs.size()) throw std::runtime_error("obj: uv index out of range"); out.uvs.push_back(uvs[2 * id.y]); out.uvs.push_back(uvs[2 * id.y + 1]); }
                else { out.uvs.push_back(0.0f); out.uvs.push_back(0.0f); } // na::origin(), obj.rs:377
                vt2id[key] = idx;
            } else idx = it->second;
            og.faces.push_back(idx);
        }
        if (!og.faces.empty()) out.groups.push_back(std::move(og));
    }
    return out;
}

// ------------------------------------------------------------------------------------ camera --
void inverse_projection(const Camera& c, double width, double height, double out16[16]) {
    double aspect = width / height, fovy = c.fovy * (3.14159265358979323846 / 180.0), zn = 1.0, zf = 100000.0;
    double t = std::tan(fovy / 2.0);
    double P[4][4] = {{1.0 / (aspect * t), 0, 0, 0}, {0, 1.0 / t, 0, 0}, {0, 0, (zf + zn) / (zn - zf), 2.0 * zf * zn / (zn - zf)}, {0, 0, -1, 0}};
    double z[3] = {c.eye[0] - c.at[0], c.eye[1] - c.at[1], c.eye[2] - c.at[2]};
    double zl = std::sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]); for (double& v : z) v /= zl;
    double up[3] = {0, 1, 0};
    double x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
    double xl = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); for (double& v : x) v /= xl;
    double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    double V[4][4] = {{x[0], x[1], x[2], 0}, {y[0], y[1], y[2], 0}, {z[0], z[1], z[2], 0}, {0, 0, 0, 1}};
    for (int r = 0; r < 3; ++r) V[r][3] = -(V[r][0] * c.eye[0] + V[r][1] * c.eye[1] + V[r][2] * c.eye[2]);
    double m[16]; // row-major P*V
    for (int r = 0; r < 4; ++r) for (int k = 0; k < 4; ++k) { double s = 0; for (int j = 0; j < 4; ++j) s += P[r][j] * V[j][k]; m[4 * r + k] = s; }
    double inv[16]; // cofactor inverse
    inv[0] = m[5]*m[10]*m[15] - m[5]*m[11]*m[14] - m[9]*m[6]*m[15] + m[9]*m[7]*m[14] + m[13]*m[6]*m[11] - m[13]*m[7]*m[10];
    inv[4] = -m[4]*m[10]*m[15] + m[4]*m[11]*m[14] + m[8]*m[6]*m[15] - m[8]*m[7]*m[14] - m[12]*m[6]*m[11] + m[12]*m[7]*m[10];
    inv[8] = m[4]*m[9]*m[15] - m[4]*m[11]*m[13] - m[8]*m[5]*m[15] + m[8]*m[7]*m[13] + m[12]*m[5]*m[11] - m[12]*m[7]*m[9];
    inv[12] = -m[4]*m[9]*m[14] + m[4]*m[10]*m[13] + m[8]*m[5]*m[14] - m[8]*m[6]*m[13] - m[12]*m[5]*m[10] + m[12]*m[6]*m[9];
    inv[1] = -m[1]*m[10]*m[15] + m[1]*m[11]*m[14] + m[9]*m[2]*m[15] - m[9]*m[3]*m[14] - m[13]*m[2]*m[11] + m[13]*m[3]*m[10];
    inv[5] = m[0]*m[10]*m[15] - m[0]*m[11]*m[14] - m[8]*m[2]*m[15] + m[8]*m[3]*m[14] + m[12]*m[2]*m[11] - m[12]*m[3]*m[10];
    inv[9] = -m[0]*m[9]*m[15] + m[0]*m[11]*m[13] + m[8]*m[1]*m[15] - m[8]*m[3]*m[13] - m[12]*m[1]*m[11] + m[12]*m[3]*m[9];
    inv[13] = m[0]*m[9]*m[14] - m[0]*m[10]*m[13] - m[8]*m[1]*m[14] + m[8]*m[2]*m[13] + m[12]*m[1]*m[10] - m[12]*m[2]*m[9];
    inv[2] = m[1]*m[6]*m[15] - m[1]*m[7]*m[14] - m[5]*m[2]*m[15] + m[5]*m[3]*m[14] + m[13]*m[2]*m[7] - m[13]*m[3]*m[6];
    inv[6] = -m[0]*m[6]*m[15] + m[0]*m[7]*m[14] + m[4]*m[2]*m[15] - m[4]*m[3]*m[14] - m[12]*m[2]*m[7] + m[12]*m[3]*m[6];
    inv[10] = m[0]*m[5]*m[15] - m[0]*m[7]*m[13] - m[4]*m[1]*m[15] + m[4]*m[3]*m[13] + m[12]*m[1]*m[7] - m[12]*m[3]*m[5];
    inv[14] = -m[0]*m[5]*m[14] + m[0]*m[6]*m[13] + m[4]*m[1]*m[14] - m[4]*m[2]*m[13] - m[12]*m[1]*m[6] + m[12]*m[2]*m[5];
    inv[3] = -m[1]*m[6]*m[11] + m[1]*m[7]*m[10] + m[5]*m[2]*m[11] - m[5]*m[3]*m[10] - m[9]*m[2]*m[7] + m[9]*m[3]*m[6];
    inv[7] = m[0]*m[6]*m[11] - m[0]*m[7]*m[10] - m[4]*m[2]*m[11] + m[4]*m[3]*m[10] + m[8]*m[2]*m[7] - m[8]*m[3]*m[6];
    inv[11] = -m[0]*m[5]*m[11] + m[0]*m[7]*m[9] + m[4]*m[1]*m[11] - m[4]*m[3]*m[9] - m[8]*m[1]*m[7] + m[8]*m[3]*m[5];
    inv[15] = m[0]*m[5]*m[10] - m[0]*m[6]*m[9] - m[4]*m[1]*m[10] + m[4]*m[2]*m[9] + m[8]*m[1]*m[6] - m[8]*m[2]*m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0) throw std::runtime_error("singular projection");
    for (int r = 0; r < 4; ++r) for (int k = 0; k < 4; ++k) out16[4 * k + r] = inv[4 * r + k] / det; // column-major out
}

// ------------------------------------------------------------------------------------ .scene --
void LoadedScene::finalize() {
    std::memset(&desc, 0, sizeof desc);
    desc.background[0] = desc.background[1] = desc.background[2] = 1.0f; // Vector3::from_element(1.0), loader3d.rs:61
    desc.num_lights = (uint32_t)lights.size(); desc.lights = lights.data();
    desc.num_materials = (uint32_t)materials.size(); desc.materials = materials.data();
    for (size_t i = 0; i < textures.size(); ++i) textures[i].texels = texture_data[i]->bytes.data();
    desc.num_textures = (uint32_t)textures.size(); desc.textures = textures.data();
    desc.num_meshes = (uint32_t)meshes.size(); desc.meshes = meshes.data();
    desc.num_nodes = (uint32_t)nodes.size(); desc.nodes = nodes.data();
}

namespace {

struct LibEntry { float alpha; int32_t material_id; };

struct Parser {
    LoadedScene& sc; std::string base; LoadOptions opt;
    std::map<std::string, LibEntry> mtllib;
    std::map<std::string, int32_t> tex_cache[2]; // loaded_opaque / loaded_transparent (texture2d.rs:31-34)

    int32_t texture(const std::string& path, bool opacity) {
        auto it = tex_cache[opacity].find(path);
        if (it != tex_cache[opacity].end()) return it->second;
        std::shared_ptr<ImageData> data;
        if (file_exists(path)) data = decode_texture(read_image(path), opacity);
        else if (opt.allow_standins && path.size() >= 9 && path.compare(path.size() - 9, 9, "globe.png") == 0 && !opacity) {
            data = globe_standin(); sc.warnings.push_back("stand-in generated for missing " + path);
        } else throw std::runtime_error("Image not found: " + path);
        NraysTexture t; std::memset(&t, 0, sizeof t);
        t.width = data->width; t.height = data->height; t.format = data->format;
        t.interp = NRAYS_INTERP_BILINEAR; t.overflow = NRAYS_OVERFLOW_WRAP; // loader3d.rs:469-473
        sc.textures.push_back(t); sc.texture_data.push_back(data);
        return tex_cache[opacity][path] = (int32_t)sc.textures.size() - 1;
    }
    int32_t phong(const float ka[3], const float kd[3], const float ks[3], int32_t tex, int32_t atex, float ns) {
        NraysMaterial m; std::memset(&m, 0, sizeof m);
        m.kind = NRAYS_MAT_PHONG;
        for (int k = 0; k < 3; ++k) { m.ambiant[k] = ka[k]; m.diffuse[k] = kd[k]; m.specular[k] = ks[k]; }
        m.shininess = ns; m.texture_id = tex; m.alpha_texture_id = atex;
        sc.materials.push_back(m);
        return (int32_t)sc.materials.size() - 1;
    }
    int32_t special(uint32_t kind) {
        NraysMaterial m; std::memset(&m, 0, sizeof m); m.kind = kind; m.texture_id = m.alpha_texture_id = -1;
        sc.materials.push_back(m); return (int32_t)sc.materials.size() - 1;
    }
    int32_t from_mtl(const MtlMaterial& m, const std::string& dir) {
        int32_t t = m.diffuse_texture.empty() ? -1 : texture(join_path(dir, m.diffuse_texture), false);
        int32_t a = m.opacity_map.empty() ? -1 : texture(join_path(dir, m.opacity_map), true);
        return phong(m.ambiant, m.diffuse, m.specular, t, a, m.shininess);
    }
};

struct Props { // examples/loader3d.rs:168-212
    size_t superbloc = 0;
    std::vector<std::pair<size_t, std::vector<std::string>>> geom; // (line, words incl. the keyword)
    bool has_pos = false, has_angle = false, has_material = false, has_eye = false, has_at = false, has_fovy = false, has_color = false,
         has_resolution = false, has_output = false;
    double pos[3], angle[3], eye[3], at[3], fovy = 0, color[3], resolution[2], refl[2] = {0, 0}, refr = 1.0, aa[2] = {1, 0}, radius = 0, nsample = 1;
    std::string material, output;
    bool solid = false;
};

} // namespace

std::unique_ptr<LoadedScene> parse_scene(const std::string& text, const std::string& base_dir, const LoadOptions& opt) {
    auto sc = std::make_unique<LoadedScene>();
    Parser P{*sc, base_dir, opt, {}, {}};
    const float w01[3] = {0.1f, 0.1f, 0.1f}, one[3] = {1, 1, 1};
    P.mtllib["normals"] = {1.0f, P.special(NRAYS_MAT_NORMAL)};       // loader3d.rs:235-248
    P.mtllib["uvs"] = {1.0f, P.special(NRAYS_MAT_UV)};
    P.mtllib["default"] = {1.0f, P.phong(w01, one, one, -1, -1, 100.0f)}; // `white`, loader3d.rs:226-233

    enum Mode { None, LightMode, ShapeMode, CameraMode } mode = None;
    Props props;
    auto need = [&](bool has, const char* what) { if (!has) fail(props.superbloc, std::string("missing attribute: ") + what); };
    auto flush = [&]() { // register(), loader3d.rs:348-362
        if (mode == LightMode) {
            need(props.has_pos, "pos <x> <y> <z>"); need(props.has_color, "color <r> <g> <b>");
            NraysLight l; std::memset(&l, 0, sizeof l);
            for (int k = 0; k < 3; ++k) { l.pos[k] = props.pos[k]; l.color[k] = (float)props.color[k]; }
            l.radius = props.radius;
            // light.rs:20 with `nsample as usize` (loader3d.rs:456): Rust's cast saturates (negative / NaN -> 0), a C++ cast
            // of such a double is undefined
            const size_t nsample = props.nsample > 0.0 ? (props.nsample < 4.0e9 ? (size_t)props.nsample : (size_t)4000000000u) : 0;
            l.racsample = (uint32_t)std::sqrt((float)nsample);
            sc->lights.push_back(l);
        } else if (mode == CameraMode) {
            need(props.has_output, "output <filename>"); need(props.has_resolution, "resolution <x> <y>");
            need(props.has_eye, "eye <x> <y> <z>"); need(props.has_at, "at <x> <y> <z>"); need(props.has_fovy, "fovy <value>");
            if (!(props.aa[0] >= 1.0)) throw std::runtime_error("The number of ray per pixel must be at least 1.0");
            Camera c;
            for (int k = 0; k < 3; ++k) { c.eye[k] = props.eye[k]; c.at[k] = props.at[k]; }
            c.fovy = props.fovy; c.resolution[0] = props.resolution[0]; c.resolution[1] = props.resolution[1];
            c.aa[0] = props.aa[0]; c.aa[1] = props.aa[1]; c.output = props.output;
            sc->cameras.push_back(c);
        } else if (mode == ShapeMode) { // register_geometry, loader3d.rs:501-792
            need(props.has_pos, "pos <x> <y> <z>"); need(props.has_angle, "angle <x> <y> <z>");
            if (props.geom.empty()) fail(props.superbloc, "missing attribute: <geom_type> <geom parameters>]");
            need(props.has_material, "material <material_name>");
            auto it = P.mtllib.find(props.material);
            if (it == P.mtllib.end()) throw std::runtime_error("Attempted to use an unknown material: " + props.material);
            bool special = props.material == "uvs" || props.material == "normals";
            NraysNode n; std::memset(&n, 0, sizeof n);
            for (int k = 0; k < 3; ++k) { n.translation[k] = props.pos[k]; n.axis_angle[k] = props.angle[k] * (3.14159265358979323846 / 180.0); }
            n.refl_mix = (float)props.refl[0]; n.refl_atenuation = (float)props.refl[1]; n.refr_coeff = props.refr;
            n.solid = props.solid ? 1u : 0u; n.alpha = it->second.alpha; n.material_id = (uint32_t)it->second.material_id; n.mesh_id = -1;
            const auto& g = props.geom[0].second; size_t gl = props.geom[0].first; // only the first geometry is used (loader3d.rs:593)
            const std::string& kind = g[0];
            if (kind == "ball") { n.shape_kind = NRAYS_SHAPE_BALL; n.params[0] = num(gl, g, 1, 1); sc->nodes.push_back(n); }
            else if (kind == "box") { n.shape_kind = NRAYS_SHAPE_CUBOID; for (int k = 0; k < 3; ++k) n.params[k] = num(gl, g, 1 + k, 3); sc->nodes.push_back(n); }
            else if (kind == "cylinder" || kind == "capsule" || kind == "cone") {
                n.shape_kind = kind == "cylinder" ? NRAYS_SHAPE_CYLINDER : kind == "capsule" ? NRAYS_SHAPE_CAPSULE : NRAYS_SHAPE_CONE;
                n.params[0] = num(gl, g, 1, 2); n.params[1] = num(gl, g, 2, 2); sc->nodes.push_back(n);
            } else if (kind == "plane") {
                double v[3] = {num(gl, g, 1, 3), num(gl, g, 2, 3), num(gl, g, 3, 3)};
                double nn = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                n.shape_kind = NRAYS_SHAPE_PLANE; for (int k = 0; k < 3; ++k) n.params[k] = v[k] / nn; // parse_plane normalises
                sc->nodes.push_back(n);
            } else if (kind == "obj") {
                if (g.size() < 3) fail(gl, "2 paths were expected, found " + std::to_string(g.size() - 1) + ".");
                std::string objpath = join_path(P.base, g[1]), mtldir = join_path(P.base, g[2]);
                ObjFile of = parse_obj_file(objpath, mtldir);
                if (!of.groups.empty()) {
                    auto verts = std::make_shared<std::vector<double>>(of.coords.size());
                    for (size_t k = 0; k < of.coords.size(); ++k) (*verts)[k] = (double)of.coords[k] / 4.0; // loader3d.rs:669
                    auto uvs = std::make_shared<std::vector<double>>(of.uvs.size());
                    for (size_t k = 0; k < of.uvs.size(); ++k) (*uvs)[k] = (double)of.uvs[k];
                    sc->vertex_arrays.push_back(verts); sc->uv_arrays.push_back(uvs);
                    for (const ObjGroup& og : of.groups) {
                        auto idx = std::make_shared<std::vector<uint32_t>>(og.faces);
                        sc->index_arrays.push_back(idx);
                        NraysMesh m; m.num_vertices = (uint32_t)(verts->size() / 3); m.num_triangles = (uint32_t)(idx->size() / 3);
                        m.vertices = verts->data(); m.uvs = uvs->data(); m.indices = idx->data();
                        sc->meshes.push_back(m);
                        NraysNode gn = n; gn.shape_kind = NRAYS_SHAPE_TRIMESH; gn.mesh_id = (int32_t)sc->meshes.size() - 1;
                        if (og.has_mtl) { // loader3d.rs:697-773
                            gn.alpha = og.mtl.alpha * it->second.alpha;
                            int32_t mid = P.from_mtl(og.mtl, mtldir);
                            if (!special) gn.material_id = (uint32_t)mid;
                        }
                        sc->nodes.push_back(gn);
                    }
                }
            } else fail(gl, "unknown geometry");
        }
    };

    std::istringstream is(text); std::string line; size_t l = 0;
    for (; std::getline(is, line); ++l) {
        auto w = words(line);
        if (w.empty() || w[0][0] == '#') continue;
        const std::string& t = w[0];
        auto trip = [&](double* out) { for (int k = 0; k < 3; ++k) out[k] = num(l, w, 1 + k, 3); };
        auto duet = [&](double* out) { for (int k = 0; k < 2; ++k) out[k] = num(l, w, 1 + k, 2); };
        if (t == "mtllib") { // register_mtllib, loader3d.rs:461-499
            std::string p = join_path(P.base, join(w, 1));
            for (const MtlMaterial& m : parse_mtl_file(p)) P.mtllib[m.name] = {m.alpha, P.from_mtl(m, P.base)};
        }
        else if (t == "light" || t == "geometry" || t == "camera") {
            flush(); props = Props(); props.superbloc = l;
            mode = t == "light" ? LightMode : t == "geometry" ? ShapeMode : CameraMode;
        }
        else if (t == "color") { trip(props.color); props.has_color = true; }
        else if (t == "angle") { trip(props.angle); props.has_angle = true; }
        else if (t == "pos") { trip(props.pos); props.has_pos = true; }
        else if (t == "eye") { trip(props.eye); props.has_eye = true; }
        else if (t == "at") { trip(props.at); props.has_at = true; }
        else if (t == "material") { props.material = join(w, 1); props.has_material = true; }
        else if (t == "fovy") { props.fovy = num(l, w, 1, 1); props.has_fovy = true; }
        else if (t == "output") { props.output = join(w, 1); props.has_output = true; }
        else if (t == "resolution") { duet(props.resolution); props.has_resolution = true; }
        else if (t == "refl") duet(props.refl);
        else if (t == "refr") props.refr = num(l, w, 1, 1);
        else if (t == "aa") duet(props.aa);
        else if (t == "radius") props.radius = num(l, w, 1, 1);
        else if (t == "nsample") props.nsample = num(l, w, 1, 1);
        else if (t == "ball" || t == "plane" || t == "box" || t == "cylinder" || t == "capsule" || t == "cone" || t == "obj") props.geom.push_back({l, w});
        else if (t == "solid") props.solid = true;
        else sc->warnings.push_back("Warning: unknown line " + std::to_string(l) + " ignored: `" + line + "'");
    }
    flush();
    sc->finalize();
    return sc;
}

std::unique_ptr<LoadedScene> load_scene_file(const std::string& path, const LoadOptions& opt) {
    return parse_scene(read_file(path), dir_of(path), opt);
}

} // namespace nrays_host
