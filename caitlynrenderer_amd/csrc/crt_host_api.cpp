// [host] entry points of include/crt.h: camera, RNG, builders, loader.  No HIP here.
#include <cstring>
#include <new>
#include <string>

#include "../../include/crt.h"
#include "crt_error.hpp"
#include "crt_handles.hpp"
#include "host/image.hpp"
#include "host/camera.hpp"
#include "host/cwbvh.hpp"
#include "host/obj_loader.hpp"
#include "host/rnd.hpp"
#include "host/sbvh.hpp"

namespace crt {
thread_local std::string g_last_error;
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
}  // namespace crt

using crt::fail;


extern "C" {

const char* crt_last_error(void) { return crt::g_last_error.c_str(); }
uint32_t crt_abi_version(void) { return CRT_ABI_VERSION; }

int crt_camera_look_at(const float pos[3], const float look_at[3], float fov_deg, crt_camera* out) {
    if (!pos || !look_at || !out) return fail(CRT_ERR_INVALID, "crt_camera_look_at: null argument");
    crt::Camera c(crt::float3(pos[0], pos[1], pos[2]), crt::float3(look_at[0], look_at[1], look_at[2]), fov_deg);
    *out = c.abi();
    return CRT_OK;
}

uint32_t crt_pcg_hash(uint32_t x) { return crt::pcg_hash(x); }
float crt_randf2(uint32_t* state) {
    crt::Rnd r;
    r.state = *state;
    float v = r.randf2();
    *state = r.state;
    return v;
}

int crt_sbvh_build(const crt_triangle* tris, size_t n_tris, const float* vertices, size_t n_vertices,
                   uint32_t flags, crt_sbvh** out) {
    if (!out) return fail(CRT_ERR_INVALID, "crt_sbvh_build: null out");
    *out = nullptr;
    if (!tris || !vertices || n_tris == 0) return fail(CRT_ERR_INVALID, "crt_sbvh_build: empty input");
    if (n_tris >= (1u << 21)) return fail(CRT_ERR_LIMIT, "crt_sbvh_build: more than 2^21 triangles (FlatNode.h:24 start field)");
    for (size_t i = 0; i < n_tris; ++i)
        for (int j = 0; j < 3; ++j)
            if (tris[i].v[j] < 0 || (size_t)tris[i].v[j] >= n_vertices)
                return fail(CRT_ERR_INVALID, "crt_sbvh_build: vertex index out of range");
    crt_sbvh* h = new (std::nothrow) crt_sbvh;
    if (!h) return fail(CRT_ERR_NOMEM, "crt_sbvh_build: out of memory");
    try {
        h->bvh.build(tris, n_tris, reinterpret_cast<const crt::float3*>(vertices), n_vertices, flags);
    } catch (const std::exception& e) {
        delete h;
        return fail(CRT_ERR_NOMEM, std::string("crt_sbvh_build: ") + e.what());
    }
    *out = h;
    return CRT_OK;
}
size_t crt_sbvh_num_nodes(const crt_sbvh* h) { return h ? h->bvh.flat_nodes.size() : 0; }
size_t crt_sbvh_num_slots(const crt_sbvh* h) { return h ? h->bvh.triangle_indices.size() : 0; }
const crt_flatnode* crt_sbvh_nodes(const crt_sbvh* h) { return h ? h->bvh.flat_nodes.data() : nullptr; }
const int32_t* crt_sbvh_triangle_indices(const crt_sbvh* h) { return h ? h->bvh.triangle_indices.data() : nullptr; }
const crt_triangle* crt_sbvh_triangles(const crt_sbvh* h) { return h ? h->bvh.triangles.data() : nullptr; }
void crt_sbvh_free(crt_sbvh* h) { delete h; }

int crt_cwbvh_convert(const crt_flatnode* bvh2, size_t n_nodes, size_t n_slots, crt_cwbvh** out) {
    if (!out) return fail(CRT_ERR_INVALID, "crt_cwbvh_convert: null out");
    *out = nullptr;
    crt_cwbvh* h = new (std::nothrow) crt_cwbvh;
    if (!h) return fail(CRT_ERR_NOMEM, "crt_cwbvh_convert: out of memory");
    bool ok = false;
    try {
        ok = h->bvh.convert(bvh2, n_nodes, n_slots, nullptr);
    } catch (const std::exception& e) {
        delete h;
        return fail(CRT_ERR_NOMEM, std::string("crt_cwbvh_convert: ") + e.what());
    }
    if (!ok) {
        std::string msg = "crt_cwbvh_convert: " + h->bvh.error;
        delete h;
        return fail(CRT_ERR_INVALID, msg);
    }
    *out = h;
    return CRT_OK;
}
size_t crt_cwbvh_num_nodes(const crt_cwbvh* h) { return h ? h->bvh.nodes.size() : 0; }
size_t crt_cwbvh_num_tris(const crt_cwbvh* h) { return h ? h->bvh.tri_slots.size() : 0; }
const crt_node8* crt_cwbvh_nodes(const crt_cwbvh* h) { return h ? h->bvh.nodes.data() : nullptr; }
const int32_t* crt_cwbvh_tri_slots(const crt_cwbvh* h) { return h ? h->bvh.tri_slots.data() : nullptr; }
const int32_t* crt_cwbvh_child_bvh2(const crt_cwbvh* h) { return h ? h->bvh.child_bvh2.data() : nullptr; }
uint32_t crt_cwbvh_depth(const crt_cwbvh* h) { return h ? h->bvh.depth : 0; }
void crt_cwbvh_free(crt_cwbvh* h) { delete h; }

int crt_load_obj(const char* path, float camera_position[3], crt_mesh** out) {
    if (!path || !out) return fail(CRT_ERR_INVALID, "crt_load_obj: null argument");
    *out = nullptr;
    crt_mesh* h = new (std::nothrow) crt_mesh;
    if (!h) return fail(CRT_ERR_NOMEM, "crt_load_obj: out of memory");
    bool ok = false;
    try {
        ok = h->mesh.read_object(path);
    } catch (const std::exception& e) {   // bad_alloc / length_error from the loader's or a texture decoder's vectors
        delete h;
        return fail(CRT_ERR_NOMEM, std::string("crt_load_obj: ") + e.what());
    }
    if (!ok) {
        std::string msg = "crt_load_obj: " + h->mesh.error;
        delete h;
        return fail(CRT_ERR_IO, msg);
    }
    if (camera_position)
        for (int k = 0; k < 3; ++k) camera_position[k] += h->mesh.translation[k];   // Scene.h:924
    *out = h;
    return CRT_OK;
}
size_t crt_mesh_counts(const crt_mesh* h, size_t* nv, size_t* nn, size_t* nt, size_t* ntri, size_t* nm, size_t* nl) {
    if (!h) return 0;
    if (nv) *nv = h->mesh.vertices.size();
    if (nn) *nn = h->mesh.normals.size();
    if (nt) *nt = h->mesh.texcoords.size() / 2;
    if (ntri) *ntri = h->mesh.triangles.size();
    if (nm) *nm = h->mesh.mats.size();
    if (nl) *nl = h->mesh.lights.size();
    return h->mesh.triangles.size();
}
const float* crt_mesh_vertices(const crt_mesh* h) { return h ? &h->mesh.vertices.data()->x : nullptr; }
const float* crt_mesh_normals(const crt_mesh* h) { return h ? &h->mesh.normals.data()->x : nullptr; }
const float* crt_mesh_texcoords(const crt_mesh* h) { return h ? h->mesh.texcoords.data() : nullptr; }
const crt_triangle* crt_mesh_triangles(const crt_mesh* h) { return h ? h->mesh.triangles.data() : nullptr; }
const crt_material* crt_mesh_materials(const crt_mesh* h) { return h ? h->mesh.mats.data() : nullptr; }
const crt_light* crt_mesh_lights(const crt_mesh* h) { return h ? h->mesh.lights.data() : nullptr; }
const float* crt_mesh_vertex_min(const crt_mesh* h) { return h ? &h->mesh.vertex_min.x : nullptr; }
const uint8_t* crt_mesh_albedo_textures(const crt_mesh* h, int32_t* width, int32_t* height, int32_t* n_layers) {
    if (width) *width = h ? h->mesh.tex_width : 0;
    if (height) *height = h ? h->mesh.tex_height : 0;
    if (n_layers) *n_layers = h ? h->mesh.n_textures : 0;
    return (h && h->mesh.n_textures > 0) ? h->mesh.albedo_textures.data() : nullptr;
}
void crt_mesh_free(crt_mesh* h) { delete h; }

int crt_image_decode(const uint8_t* file_bytes, size_t n_bytes, int32_t* width, int32_t* height, uint8_t* rgb, size_t rgb_capacity) {
    if (!file_bytes || !width || !height) return fail(CRT_ERR_INVALID, "crt_image_decode: null argument");
    int w = 0, h = 0;
    std::vector<uint8_t> px;
    std::string err;
    try {
        if (!crt::decode_image_rgb8(file_bytes, n_bytes, w, h, px, err)) return fail(CRT_ERR_INVALID, "crt_image_decode: " + err);
    } catch (const std::exception& e) {
        return fail(CRT_ERR_NOMEM, std::string("crt_image_decode: ") + e.what());
    }
    *width = w; *height = h;
    if (rgb) {
        if (rgb_capacity < px.size()) return fail(CRT_ERR_INVALID, "crt_image_decode: output buffer too small");
        std::memcpy(rgb, px.data(), px.size());
    }
    return CRT_OK;
}
int crt_image_encode_png(const uint8_t* pixels, int32_t width, int32_t height, int32_t channels, int32_t bottom_up,
                         uint8_t* file, size_t file_capacity, size_t* file_size) {
    if (!pixels || !file_size || width <= 0 || height <= 0 || (channels != 3 && channels != 4))
        return fail(CRT_ERR_INVALID, "crt_image_encode_png: bad argument");
    try {
        std::vector<uint8_t> out;
        crt::encode_png(pixels, width, height, channels, bottom_up != 0, out);
        *file_size = out.size();
        if (file) {
            if (file_capacity < out.size()) return fail(CRT_ERR_INVALID, "crt_image_encode_png: output buffer too small");
            std::memcpy(file, out.data(), out.size());
        }
    } catch (const std::exception& e) {
        return fail(CRT_ERR_NOMEM, std::string("crt_image_encode_png: ") + e.what());
    }
    return CRT_OK;
}
int crt_texture_to_array_bytes(const uint8_t* rgb, int32_t width, int32_t height, int32_t out_w, int32_t out_h, uint8_t* out) {
    if (!rgb || !out || width <= 0 || height <= 0 || out_w <= 0 || out_h <= 0) return fail(CRT_ERR_INVALID, "crt_texture_to_array_bytes: bad argument");
    try {
        crt::texture_to_array_bytes(rgb, width, height, out_w, out_h, out);
    } catch (const std::exception& e) {
        return fail(CRT_ERR_NOMEM, std::string("crt_texture_to_array_bytes: ") + e.what());
    }
    return CRT_OK;
}

}  // extern "C"
