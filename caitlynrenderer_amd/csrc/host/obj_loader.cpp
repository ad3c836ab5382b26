#include "obj_loader.hpp"
#include "image.hpp"

#include <climits>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace crt {
namespace {

inline int fix_index(int v, int n) { return v < 0 ? v + n : v > 0 ? v - 1 : -1; }   // Scene.h:135-138
inline int trunc_i32(float f) { return (!(f > -2147483648.0f && f < 2147483648.0f)) ? INT_MIN : (int)f; }
inline const char* skip_ws(const char* t) { return t + std::strspn(t, " \t"); }

struct Corner { int v = 0, vt = 0, vn = 0; };

// Scene.h:186-315.  Accepted corner forms: v/vt/vn, v/vt, v//vn.  A bare `v` corner
// matches no branch of the reference and yields no triangles; kept.  Runs of blanks are
// one separator here (the reference would read a 0 index out of each extra blank).
// Returns the number of data per corner (1..3) and whether the `//` form was seen.
int parse_corners(const char* t, std::vector<Corner>& out, bool& double_slash) {
    double_slash = false;
    int per_corner = 0;
    while (*t) {
        t = skip_ws(t);
        if (!(*t == '-' || (*t >= '0' && *t <= '9'))) break;
        int vals[3] = {0, 0, 0};
        int n = 0;
        for (;;) {
            int sign = 1;
            long long acc = 0;
            if (*t == '-') { sign = -1; ++t; }
            while (*t >= '0' && *t <= '9') { acc = 10 * acc + (*t++ - '0'); if (acc > 0x7fffffffLL) acc = 0x7fffffffLL; }   // saturates: out of range either way
            if (n < 3) vals[n] = sign * (int)acc;
            ++n;
            if (*t != '/') break;
            ++t;
            if (*t == '/') { double_slash = true; ++t; }
        }
        if (out.empty()) per_corner = n > 3 ? 3 : n;
        Corner c;
        c.v = vals[0];
        if (per_corner == 3) { c.vt = vals[1]; c.vn = vals[2]; }
        else if (per_corner == 2) { if (double_slash) c.vn = vals[1]; else c.vt = vals[1]; }
        out.push_back(c);
    }
    return per_corner;
}

}  // namespace

bool Mesh::read_mtl(const std::string& file_name, const std::string& directory, std::vector<std::pair<std::string, int>>& mtl_map) {
    std::ifstream f(file_name);
    if (!f) { error = "mtl file not found: " + file_name; return false; }   // Scene.h:510-511 prints and goes on
    std::vector<std::string> lines;
    for (std::string line; std::getline(f, line);) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        lines.push_back(line);
    }
    size_t n_materials = 0;
    for (const std::string& l : lines) n_materials += (!l.empty() && l[0] == 'n');   // Scene.h:522-523
    crt_material blank;
    for (int i = 0; i < 4; ++i) { blank.albedo[i] = 0.f; blank.emission[i] = -1.f; blank.specular[i] = 0.f; blank.tex_ind[i] = -1.f; }
    mats.assign(n_materials, blank);

    int cur = -1, n_lights = 0;
    char name[256];
    std::vector<std::string> texture_names;                            // Scene.h:555 texture_map, in order of first use
    albedo_textures.clear();
    n_textures = 0;
    for (const std::string& l : lines) {
        const char* t = skip_ws(l.c_str());
        if (std::strncmp(t, "map_Kd", 6) == 0 && cur >= 0 && (size_t)cur < mats.size()) {   // Scene.h:597-677
            if (std::sscanf(t + 6, "%255s", name) != 1) continue;
            std::string real_name = name;                              // getfilename, Scene.h:179-184: the multi-character
            const size_t bs = real_name.find_last_of('\\');             // literal '/\\' it searches for is a backslash
            if (bs != std::string::npos) real_name = real_name.substr(bs + 1);
            bool seen = false;
            for (const std::string& n : texture_names) seen = seen || n == real_name;
            if (seen) continue;                                        // Scene.h:604: a texture's second user keeps tex_ind = -1
            int w = 0, h = 0;
            std::vector<uint8_t> rgb;
            std::string err;
            if (!decode_image_rgb8(directory + real_name, w, h, rgb, err)) { error = err; return false; }
            mats[cur].tex_ind[0] = (float)texture_names.size();
            texture_names.push_back(real_name);
            const size_t layer = (size_t)tex_width * tex_height * 3;
            albedo_textures.resize(albedo_textures.size() + layer);
            texture_to_array_bytes(rgb.data(), w, h, tex_width, tex_height, albedo_textures.data() + albedo_textures.size() - layer);
            n_textures = (int)texture_names.size();
            continue;
        }
        if (std::strncmp(t, "newmtl", 6) == 0) {                       // Scene.h:567-575
            if (std::sscanf(t + 6, "%255s", name) == 1) mtl_map.emplace_back(name, ++cur);
        } else if (cur < 0 || (size_t)cur >= mats.size()) {
            continue;
        } else if (std::strncmp(t, "type", 4) == 0) {                  // Scene.h:576-582
            if (std::sscanf(t + 4, "%255s", name) == 1) {
                if (std::strncmp(name, "Mirror", 6) == 0) mats[cur].albedo[3] = 1.0f;         // Mirror_type, Scene.h:114
                // extension (no reference code, DESIGN.md "materials"): the enum's Disney_type (Scene.h:131) for the
                // README's Disney BSDF; its parameters come from the PBR lines `Pm` / `Pr` below
                else if (std::strncmp(name, "Disney", 6) == 0) mats[cur].albedo[3] = 17.0f;
            }
        } else if (t[0] == 'P' && (t[1] == 'm' || t[1] == 'r') && (t[2] == ' ' || t[2] == '\t')) {
            // extension: `Pm metallic` / `Pr roughness` (the usual PBR additions to .mtl) -> specular.x / specular.y,
            // read by the Disney lobe only; the reference ignores these lines (no branch matches them)
            float v = 0.f;
            if (std::sscanf(t + 2, "%f", &v) == 1) mats[cur].specular[t[1] == 'm' ? 0 : 1] = v;
        } else if (t[0] == 'K') {                                      // Scene.h:583-596
            float e[4] = {-1.f, -1.f, -1.f, -1.f};
            if (t[1] == 'd') std::sscanf(t + 2, "%f %f %f", &mats[cur].albedo[0], &mats[cur].albedo[1], &mats[cur].albedo[2]);
            if (t[1] == 'e') std::sscanf(t + 2, "%f %f %f", &e[0], &e[1], &e[2]);
            if (e[0] > 0 || e[1] > 0 || e[2] > 0) e[3] = (float)n_lights++;
            for (int i = 0; i < 4; ++i) mats[cur].emission[i] = e[i];   // every K? line overwrites it
        }
    }
    return true;
}

bool Mesh::read_object(const std::string& file_name) {
    std::ifstream f(file_name);
    if (!f) { error = "obj file not found: " + file_name; return false; }
    const size_t slash = file_name.find_last_of("/\\");
    const std::string dir = slash == std::string::npos ? std::string() : file_name.substr(0, slash + 1);

    std::vector<std::pair<std::string, int>> mtl_map;
    bool read_mtl_done = false;
    int mtl_ind = 0;
    float3 vmin(1e20f), vmax(-1e20f);
    char name[256];
    std::vector<Corner> corners;

    for (std::string line; std::getline(f, line);) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const char* t = skip_ws(line.c_str());
        if (t[0] == 'v') {                                             // Scene.h:776-814
            float x = 0.f, y = 0.f, z = 0.f;
            if (t[1] == ' ' || t[1] == '\t') {
                std::sscanf(t + 1, "%f %f %f", &x, &y, &z);
                float3 p(x, y, z);
                vmin = {fmin_(vmin.x, p.x), fmin_(vmin.y, p.y), fmin_(vmin.z, p.z)};
                vmax = {fmax_(vmax.x, p.x), fmax_(vmax.y, p.y), fmax_(vmax.z, p.z)};
                vertices.push_back(p);
            } else if (t[1] == 't') {
                std::sscanf(t + 2, "%f %f", &x, &y);
                texcoords.push_back(x);
                texcoords.push_back(1.0f - y);
            } else if (t[1] == 'n') {
                std::sscanf(t + 2, "%f %f %f", &x, &y, &z);
                normals.emplace_back(x, y, z);
            }
        } else if (t[0] == 'f') {                                      // Scene.h:815-882
            corners.clear();
            bool dbl = false;
            const int per = parse_corners(t + 1, corners, dbl);
            if (per < 2 || corners.size() < 3) continue;
            const int nv = (int)vertices.size(), nvt = (int)(texcoords.size() / 2), nvn = (int)normals.size();
            for (Corner& c : corners) {
                c.v = fix_index(c.v, nv);
                c.vt = (per == 3 || !dbl) ? fix_index(c.vt, nvt) : -1;
                c.vn = (per == 3 || dbl) ? fix_index(c.vn, nvn) : -1;
            }
            for (size_t i = 0; i + 2 < corners.size(); ++i) {          // fan (0, i+1, i+2)
                const Corner &a = corners[0], &b = corners[i + 1], &c = corners[i + 2];
                if (a.v < 0 || b.v < 0 || c.v < 0 || a.v >= nv || b.v >= nv || c.v >= nv) {
                    error = "face references a vertex that does not exist"; return false;
                }
                crt_triangle tr;
                tr.v[0] = a.v; tr.v[1] = b.v; tr.v[2] = c.v; tr.v[3] = mtl_ind;
                tr.vt[0] = a.vt; tr.vt[1] = b.vt; tr.vt[2] = c.vt; tr.vt[3] = 0;
                const float3 p0 = vertices[a.v], p1 = vertices[b.v], p2 = vertices[c.v];
                if (a.vn == -1) {                                      // Scene.h:843-853: integer-truncated normal
                    const float3 n = cross(p1 - p0, p2 - p0);
                    tr.vn[0] = trunc_i32(n.x); tr.vn[1] = trunc_i32(n.y); tr.vn[2] = trunc_i32(n.z); tr.vn[3] = 0;
                } else {
                    tr.vn[0] = a.vn; tr.vn[1] = b.vn; tr.vn[2] = c.vn; tr.vn[3] = 1;
                }
                if ((size_t)mtl_ind < mats.size() && mats[mtl_ind].emission[3] != -1.0f) {   // Scene.h:856-878
                    const float3 u = p1 - p0, v = p2 - p0;
                    float3 n = cross(u, v);
                    const float area = length(n);
                    n = normalize(n);
                    crt_light L;
                    for (int k = 0; k < 3; ++k) {
                        L.p[k] = p0[k]; L.u[k] = u[k]; L.v[k] = v[k]; L.n[k] = n[k];
                        L.e[k] = mats[mtl_ind].emission[k];
                    }
                    L.area_pdf[0] = area; L.area_pdf[1] = 0.f; L.area_pdf[2] = 0.f;
                    lights.push_back(L);
                }
                triangles.push_back(tr);
            }
        } else if (std::strncmp(t, "usemtl", 6) == 0) {                // Scene.h:883-889
            if (std::sscanf(t + 6, "%255s", name) == 1) {
                mtl_ind = 0;   // an unknown name maps to 0 (operator[] default-inserts, Scene.h:888)
                for (const auto& kv : mtl_map) if (kv.first == name) mtl_ind = kv.second;
            }
        } else if (t[0] == 'm' && !read_mtl_done) {                    // Scene.h:890-899 (first `m…` line = mtllib)
            // the reference skips 6 characters of whatever `m…` line comes first; a line shorter than that has no name
            if (std::strlen(t) > 6 && std::sscanf(t + 6, "%255s", name) == 1) {
                if (!read_mtl(dir + name, dir, mtl_map)) return false;
                read_mtl_done = true;
            }
        }
    }

    float sum_area = 0.f;                                              // Scene.h:904-913
    for (const crt_light& L : lights) sum_area += L.area_pdf[0];
    if (sum_area > 0.f) {
        const float inv = 1.0f / sum_area;
        for (crt_light& L : lights) L.area_pdf[1] = L.area_pdf[0] * inv;
    }
    vertex_min = vmin;                                                 // Scene.h:915-925
    translation = -vmin;
    for (float3& v : vertices) v += translation;
    for (crt_light& L : lights) for (int k = 0; k < 3; ++k) L.p[k] += translation[k];
    return true;
}

}  // namespace crt
