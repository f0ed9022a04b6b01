// pt_obj.cpp — host-side OBJ/MTL loading, the stand-in for obj.rs:13-107 (`load_obj`) on top of
// tobj 0.1.6.  tobj is not vendored in the reference tree and has no tests there ("parity
// unpinned"); the semantics restated here are the ones its published 0.1.x source
// documents and the reference's own model files exercise:
//   * a model ends at `o`/`g`, or at a `usemtl` that changes the material after faces were read;
//     the last model is pushed at end of file; default name "unnamed_object";
//   * per-model re-indexing: each distinct (v, vt, vn) triple becomes one vertex, so positions and
//     texcoords share an index (triangle_mesh.rs:86-91);
//   * quads -> (0,1,2),(0,2,3); polygons -> fans; points/lines ignored; negative indices are
//     relative to the elements read so far; `vn` is parsed and otherwise unused by the reference;
//   * `mtllib` is resolved next to the OBJ; a `usemtl` naming an unknown material yields None;
//   * MTL: `illum` is optional (Option<u8>); obj.rs:59-76 gives a model without material id or
//     without `illum` the black non-emissive default Lambertian.
// Flattening: models are appended in file order into one vertex/triangle list (no instancing).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <new>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "pt_host.h"

using pth::set_error;

namespace {

struct MtlEntry {
    std::string name;
    float ambient[3] = {0, 0, 0}, diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0};
    float shininess = 0.f;
    float optical_density = 0.f;  // Ni
    std::string diffuse_texture;
    bool has_illum = false;
    int illum = 0;
};

struct VIdx {
    long v = -1, vt = -1, vn = -1;
    bool operator<(const VIdx& o) const { return std::tie(v, vt, vn) < std::tie(o.v, o.vt, o.vn); }
};

std::vector<std::string> split_ws(const std::string& line) {
    std::vector<std::string> out;
    std::istringstream is(line);
    std::string w;
    while (is >> w) out.push_back(w);
    return out;
}

std::string rest_of_line(const std::string& line, const std::string& first) {
    size_t p = line.find(first);
    std::string r = line.substr(p + first.size());
    size_t a = r.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = r.find_last_not_of(" \t\r\n");
    return r.substr(a, b - a + 1);
}

bool parse_floats(const std::vector<std::string>& w, size_t n, float* out) {
    for (size_t i = 0; i < n; ++i) {
        if (1 + i >= w.size()) return false;
        char* end = nullptr;
        out[i] = std::strtof(w[1 + i].c_str(), &end);
        if (end == w[1 + i].c_str()) return false;
    }
    return true;
}

bool load_mtl(const std::string& path, std::vector<MtlEntry>& mats, std::map<std::string, int>& by_name) {
    std::ifstream f(path);
    if (!f) return false;
    MtlEntry cur;
    std::string line;
    auto flush = [&]() {
        if (!cur.name.empty()) {
            by_name[cur.name] = (int)mats.size();
            mats.push_back(cur);
        }
    };
    while (std::getline(f, line)) {
        auto w = split_ws(line);
        if (w.empty() || w[0] == "#") continue;
        const std::string& k = w[0];
        if (k == "newmtl") {
            flush();
            cur = MtlEntry();
            cur.name = rest_of_line(line, "newmtl");
        } else if (k == "Ka") parse_floats(w, 3, cur.ambient);
        else if (k == "Kd") parse_floats(w, 3, cur.diffuse);
        else if (k == "Ks") parse_floats(w, 3, cur.specular);
        else if (k == "Ns") parse_floats(w, 1, &cur.shininess);
        else if (k == "Ni") parse_floats(w, 1, &cur.optical_density);
        else if (k == "map_Kd") cur.diffuse_texture = rest_of_line(line, "map_Kd");
        else if (k == "illum") {
            if (w.size() > 1) { cur.has_illum = true; cur.illum = std::atoi(w[1].c_str()); }
        }
        // everything else lands in tobj's unknown_param map and is unused by the reference
    }
    flush();
    return true;
}

struct ModelOut {
    std::string name;
    uint32_t first_tri = 0, n_tris = 0;
    int material = -1;  // tobj material id or -1 (None)
    bool has_uv = false;
};

std::string dir_of(const std::string& path) {
    size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string(".") : path.substr(0, p);
}
std::string base_of(const std::string& path) {
    size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? path : path.substr(p + 1);
}
bool file_exists(const std::string& p) {
    FILE* f = std::fopen(p.c_str(), "rb");
    if (!f) return false;
    std::fclose(f);
    return true;
}

}  // namespace

struct pt_model {
    std::vector<float> verts;
    std::vector<uint32_t> tris;
    std::vector<float> tri_uvs;
    std::vector<uint32_t> tri_material;
    std::vector<pt_material> mats;
    std::vector<pt_texture> texs;
    std::vector<std::vector<uint8_t>> tex_pixels;
    std::vector<ModelOut> models;
    pt_scene_desc desc;
};

static int load_obj_impl(const char* obj_path, uint32_t flags, pt_model** out);

extern "C" {

int pt_load_obj(const char* obj_path, uint32_t flags, pt_model** out) {
    if (!obj_path || !out) { set_error("null argument"); return PT_ERR_INVALID; }
    *out = nullptr;
    try {
        return load_obj_impl(obj_path, flags, out);
    } catch (const std::bad_alloc&) {  // nothing may unwind across the C ABI (a huge or corrupt file)
        set_error("%s: out of memory while loading", obj_path);
        return PT_ERR_NOMEM;
    }
}

}  // extern "C"

static int load_obj_impl(const char* obj_path, uint32_t flags, pt_model** out) {
    std::ifstream f(obj_path);
    if (!f) { set_error("Failed to load %s", obj_path); return PT_ERR_IO; }  // obj.rs:16 panics
    const std::string dir = dir_of(obj_path);

    std::vector<float> pos, tex;  // tmp_pos / tmp_texcoord
    size_t n_normals = 0;
    std::vector<std::vector<VIdx>> faces;  // tmp_faces of the model being read
    std::vector<MtlEntry> mtl;
    std::map<std::string, int> mtl_by_name;
    std::string name = "unnamed_object";
    int mat_id = -1;

    std::unique_ptr<pt_model> M(new pt_model());
    bool any_uv = false;

    auto export_faces = [&]() {
        ModelOut mo;
        mo.name = name;
        mo.material = mat_id;
        mo.first_tri = (uint32_t)(M->tris.size() / 3);
        std::map<VIdx, uint32_t> index_map;
        std::vector<float> mpos, muv;
        std::vector<uint32_t> idx;
        auto add_vertex = [&](const VIdx& v) {
            auto it = index_map.find(v);
            if (it != index_map.end()) { idx.push_back(it->second); return; }
            uint32_t id = (uint32_t)(mpos.size() / 3);
            for (int a = 0; a < 3; ++a) mpos.push_back(pos[(size_t)v.v * 3 + a]);
            if (!tex.empty() && v.vt >= 0) { muv.push_back(tex[(size_t)v.vt * 2]); muv.push_back(tex[(size_t)v.vt * 2 + 1]); }
            index_map[v] = id;
            idx.push_back(id);
        };
        for (auto& fc : faces) {
            if (fc.size() < 3) continue;  // points and lines carry no triangles
            for (size_t k = 2; k < fc.size(); ++k) {  // triangle, quad (0,1,2),(0,2,3), polygon fan
                add_vertex(fc[0]);
                add_vertex(fc[k - 1]);
                add_vertex(fc[k]);
            }
        }
        mo.has_uv = !muv.empty();
        uint32_t vbase = (uint32_t)(M->verts.size() / 3);
        M->verts.insert(M->verts.end(), mpos.begin(), mpos.end());
        for (size_t t = 0; t + 2 < idx.size(); t += 3) {
            float uv6[6] = {0.f, 0.f, 1.f, 0.f, 1.f, 1.f};  // triangle_mesh.rs:93
            for (int k = 0; k < 3; ++k) {
                M->tris.push_back(vbase + idx[t + k]);
                if (mo.has_uv && (size_t)idx[t + k] * 2 + 1 < muv.size()) {
                    uv6[2 * k] = muv[(size_t)idx[t + k] * 2];
                    uv6[2 * k + 1] = muv[(size_t)idx[t + k] * 2 + 1];
                }
            }
            M->tri_uvs.insert(M->tri_uvs.end(), uv6, uv6 + 6);
        }
        mo.n_tris = (uint32_t)(M->tris.size() / 3) - mo.first_tri;
        any_uv = any_uv || mo.has_uv;
        M->models.push_back(mo);
    };

    std::string line;
    size_t lineno = 0;
    while (std::getline(f, line)) {
        ++lineno;
        auto w = split_ws(line);
        if (w.empty() || w[0] == "#") continue;
        const std::string& k = w[0];
        if (k == "v") {
            float v[3];
            if (!parse_floats(w, 3, v)) { set_error("%s:%zu: bad vertex", obj_path, lineno); return PT_ERR_IO; }
            pos.insert(pos.end(), v, v + 3);
        } else if (k == "vt") {
            float v[2];
            if (!parse_floats(w, 2, v)) { set_error("%s:%zu: bad texcoord", obj_path, lineno); return PT_ERR_IO; }
            tex.insert(tex.end(), v, v + 2);
        } else if (k == "vn") {
            ++n_normals;
        } else if (k == "f") {
            std::vector<VIdx> fc;
            for (size_t i = 1; i < w.size(); ++i) {
                VIdx vi;
                long vals[3] = {0, 0, 0};
                bool present[3] = {false, false, false};
                size_t start = 0;
                for (int part = 0; part < 3 && start <= w[i].size(); ++part) {
                    size_t slash = w[i].find('/', start);
                    std::string tok = w[i].substr(start, slash == std::string::npos ? std::string::npos : slash - start);
                    if (!tok.empty()) { vals[part] = std::strtol(tok.c_str(), nullptr, 10); present[part] = true; }
                    if (slash == std::string::npos) break;
                    start = slash + 1;
                }
                const long counts[3] = {(long)(pos.size() / 3), (long)(tex.size() / 2), (long)n_normals};
                long res[3] = {-1, -1, -1};
                for (int part = 0; part < 3; ++part) {
                    if (!present[part]) continue;
                    long x = vals[part];
                    res[part] = x < 0 ? counts[part] + x : x - 1;
                }
                if (!present[0] || res[0] < 0 || res[0] >= counts[0] || (present[1] && (res[1] < 0 || res[1] >= counts[1]))) {
                    set_error("%s:%zu: face index out of range", obj_path, lineno);
                    return PT_ERR_IO;
                }
                vi.v = res[0]; vi.vt = res[1]; vi.vn = res[2];
                fc.push_back(vi);
            }
            faces.push_back(fc);
        } else if (k == "o" || k == "g") {
            if (!faces.empty()) { export_faces(); faces.clear(); }
            name = rest_of_line(line, k);
            if (name.empty()) name = "unnamed_object";
        } else if (k == "mtllib") {
            std::string lib = rest_of_line(line, "mtllib");
            std::vector<MtlEntry> more;
            std::map<std::string, int> more_by_name;
            if (!load_mtl(dir + "/" + lib, more, more_by_name)) {
                set_error("%s:%zu: cannot read material library %s", obj_path, lineno, lib.c_str());
                return PT_ERR_IO;
            }
            int offset = (int)mtl.size();
            mtl.insert(mtl.end(), more.begin(), more.end());
            for (auto& kv : more_by_name) mtl_by_name[kv.first] = kv.second + offset;
        } else if (k == "usemtl") {
            std::string mname = rest_of_line(line, "usemtl");
            if (mname.empty()) { set_error("%s:%zu: usemtl without a name", obj_path, lineno); return PT_ERR_IO; }
            auto it = mtl_by_name.find(mname);
            int new_mat = it == mtl_by_name.end() ? -1 : it->second;
            if (new_mat != mat_id && !faces.empty()) { export_faces(); faces.clear(); }
            mat_id = new_mat;
        }
    }
    export_faces();  // the last model is always pushed

    // ---- materials: one per MTL entry + the default at the end (obj.rs:55-76)
    const bool want_glossy = (flags & PT_LOAD_GLOSSY) != 0, want_tex = (flags & PT_LOAD_TEXTURES) != 0;
    std::map<std::string, int> tex_by_path;
    for (auto& m : mtl) {
        pt_material pm;
        std::memset(&pm, 0, sizeof pm);
        pm.kind = PT_MAT_LAMBERT;
        pm.tex = -1;
        if (m.has_illum) {
            for (int c = 0; c < 3; ++c) { pm.color[c] = m.diffuse[c]; pm.emission[c] = m.ambient[c]; }
            pm.roughness = m.shininess;
            if (want_glossy && m.illum == 2) {  // obj.rs:38-40 (commented wiring): Glossy::new(Ns, _, Kd)
                pm.kind = (flags & PT_LOAD_GGX) ? PT_MAT_GLOSSY_GGX : PT_MAT_GLOSSY;
                for (int c = 0; c < 3; ++c) pm.emission[c] = 0.f;
            } else if ((flags & PT_LOAD_SPECULAR) && (m.illum == 3 || m.illum == 4)) {
                // IlluminationModel::from (material.rs:157-170): 3 = Mirror, 4 = Transparent
                pm.kind = m.illum == 3 ? PT_MAT_MIRROR : PT_MAT_TRANSPARENT;
                pm.ior = m.optical_density > 0.f ? m.optical_density : 1.5f;  // Material::glass(), material.rs:117
                for (int c = 0; c < 3; ++c) pm.emission[c] = 0.f;
            } else if (want_tex && !m.diffuse_texture.empty()) {  // obj.rs:41-47
                std::string cand[3] = {m.diffuse_texture, dir + "/" + m.diffuse_texture, dir + "/" + base_of(m.diffuse_texture)};
                std::string found;
                for (auto& c : cand) if (file_exists(c)) { found = c; break; }
                if (found.empty()) { set_error("Couldn't find file: %s", m.diffuse_texture.c_str()); return PT_ERR_IO; }  // mipmap.rs:61
                auto it = tex_by_path.find(found);
                if (it == tex_by_path.end()) {
                    uint32_t tw = 0, th = 0;
                    std::vector<uint8_t> px;
                    int rc = pth::decode_png(found.c_str(), tw, th, px);
                    if (rc != PT_OK) return rc;
                    pt_texture t;
                    t.width = tw; t.height = th; t.rgb8 = nullptr;
                    M->tex_pixels.push_back(std::move(px));
                    M->texs.push_back(t);
                    it = tex_by_path.emplace(found, (int)M->texs.size() - 1).first;
                }
                pm.tex = it->second;
            }
        }
        M->mats.push_back(pm);
    }
    for (size_t i = 0; i < M->texs.size(); ++i) M->texs[i].rgb8 = M->tex_pixels[i].data();
    pt_material def;
    std::memset(&def, 0, sizeof def);
    def.kind = PT_MAT_LAMBERT;
    def.tex = -1;
    M->mats.push_back(def);
    const uint32_t default_mat = (uint32_t)M->mats.size() - 1;
    M->tri_material.resize(M->tris.size() / 3);
    for (auto& mo : M->models)
        for (uint32_t t = mo.first_tri; t < mo.first_tri + mo.n_tris; ++t)
            M->tri_material[t] = mo.material >= 0 ? (uint32_t)mo.material : default_mat;

    pt_scene_desc& d = M->desc;
    std::memset(&d, 0, sizeof d);
    d.verts = M->verts.data();
    d.n_verts = (uint32_t)(M->verts.size() / 3);
    d.tris = M->tris.data();
    d.n_tris = (uint32_t)(M->tris.size() / 3);
    d.tri_uvs = any_uv ? M->tri_uvs.data() : nullptr;
    d.tri_material = M->tri_material.data();
    d.mats = M->mats.data();
    d.n_mats = (uint32_t)M->mats.size();
    d.texs = M->texs.empty() ? nullptr : M->texs.data();
    d.n_texs = (uint32_t)M->texs.size();
    *out = M.release();
    return PT_OK;
}

extern "C" {

const pt_scene_desc* pt_model_desc(const pt_model* m) { return m ? &m->desc : nullptr; }
uint32_t pt_model_count(const pt_model* m) { return m ? (uint32_t)m->models.size() : 0; }
const char* pt_model_name(const pt_model* m, uint32_t i) { return (m && i < m->models.size()) ? m->models[i].name.c_str() : ""; }
uint32_t pt_model_first_tri(const pt_model* m, uint32_t i) {
    if (!m) return 0;
    return i < m->models.size() ? m->models[i].first_tri : (uint32_t)(m->tris.size() / 3);
}
void pt_model_destroy(pt_model* m) { delete m; }

}  // extern "C"
