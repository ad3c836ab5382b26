// Host-side `Scene` with the reference's shape (Caitlyn/Scene.h:380-1249), on top of the C ABI.
//
// Same life cycle and method names as the reference class so that its callers (main.cpp:244 init,
// :297-299 per frame) read the same: construct from an OBJ path, `update()` after camera motion,
// `Render()` once per frame.  Where the reference uploads texture buffers and draws a quad, this class
// calls crt_scene_create / crt_set_camera / crt_render_frame; the display pass (output.fs) is
// `resolve()`.  Header-only; link against libcrt.so.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../../include/crt.h"
#include "camera.hpp"
#include "cwbvh.hpp"
#include "obj_loader.hpp"
#include "rnd.hpp"
#include "sbvh.hpp"

namespace crt {

struct Scene {
    // ---- CPU data (Scene.h:384-422)
    int frame_count = 0;
    Camera camera;
    std::vector<float3> vertices, normals;
    std::vector<float> texcoords;
    std::vector<crt_triangle> triangles;            // BVH leaf order after build_bvh (sbvh.h:130-139)
    std::vector<int32_t> triangle_indices;          // leaf slot -> original triangle (sbvh.h:85)
    std::vector<crt_material> mats;
    std::vector<crt_light> lights;
    std::vector<crt_flatnode> flat_nodes;
    std::vector<uint8_t> albedo_textures_data;      // Scene.h:408: RGB8 layers of tex_height x tex_width
    int tex_width = 0, tex_height = 0, n_textures = 0;
    Rnd rnd;                                        // Rnd.h:7 (thread_local there; one caller thread here too)
    uint32_t width = 700, height = 700, max_depth = 3;   // Scene.h:37, path_trace.fs:867
    std::string error;

    // ---- device handle (replaces the GLuint block, Scene.h:425-442)
    crt_scene* gpu = nullptr;

    Scene() = default;
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    // Scene(file_name, shader_direction), Scene.h:447-505.  The shader directory has no meaning here; the
    // resolution, hard-coded to 700x700 in the reference (Scene.h:37), is a parameter.
    Scene(const std::string& file_name, uint32_t w, uint32_t h, uint32_t depth = 3) : width(w), height(h), max_depth(depth) {
        camera = Camera(float3(-2.755610f, 2.745992f, 7.58545f), float3(-2.755610f, 2.745992f, 6.58545f), 40.0f);   // Scene.h:468
        if (!Read_Object(file_name)) return;
        build_bvh();
        gpu_data();
        delete_cpu_data();                          // Scene.h:503
    }
    ~Scene() { delete_gpu_data(); }

    bool ok() const { return gpu != nullptr && error.empty(); }

    bool Read_Object(const std::string& file_name) {   // Scene.h:742-926
        Mesh m;
        if (!m.read_object(file_name)) { error = m.error; std::printf("%s\n", error.c_str()); return false; }   // prints and goes on, Scene.h:746-747
        vertices = std::move(m.vertices); normals = std::move(m.normals); texcoords = std::move(m.texcoords);
        triangles = std::move(m.triangles); mats = std::move(m.mats); lights = std::move(m.lights);
        albedo_textures_data = std::move(m.albedo_textures);
        tex_width = m.tex_width; tex_height = m.tex_height; n_textures = m.n_textures;
        camera.position += m.translation;           // Scene.h:924
        return true;
    }

    void build_bvh() {                               // Scene.h:929-959
        SBVH sbvh(triangles, vertices);
        flat_nodes = sbvh.flat_nodes;                // Scene.h:943
        triangles = sbvh.triangles;                  // the reorder SBVH does in place on its argument (sbvh.h:139)
        triangle_indices = sbvh.triangle_indices;
    }

    void gpu_data() {                                // Scene.h:1000-1156
        crt_scene_desc d{};
        d.abi_version = CRT_ABI_VERSION;
        d.vertices = vertices.empty() ? nullptr : &vertices[0].x; d.n_vertices = vertices.size();
        d.normals = normals.empty() ? nullptr : &normals[0].x;    d.n_normals = normals.size();
        d.texcoords = texcoords.data();                           d.n_texcoords = texcoords.size() / 2;
        d.triangles = triangles.data();                           d.n_triangles = triangles.size();
        d.tri_orig_ids = triangle_indices.empty() ? nullptr : triangle_indices.data();
        d.materials = mats.data();                                d.n_materials = mats.size();
        d.lights = lights.data();                                 d.n_lights = lights.size();
        d.bvh = flat_nodes.data();                                d.n_bvh = flat_nodes.size();
        if (n_textures > 0) {                                     // Scene.h:1065-1078
            d.albedo_textures = albedo_textures_data.data();
            d.tex_width = tex_width; d.tex_height = tex_height; d.n_textures = n_textures;
        }
        d.width = width; d.height = height; d.max_depth = max_depth;
        if (crt_scene_create(&d, &gpu) != CRT_OK) { error = crt_last_error(); std::printf("%s\n", error.c_str()); gpu = nullptr; return; }
        // this class is the reference's frame loop as a host would write it: let the library deal the frame's tiles to as many streams
        // of the GPU as its measurements favour (3 for a scene of a few nodes, 2 for max_depth > 1 — the shipped Cornell box at depth 3
        // renders 14 % faster —, else 1); same sums either way
        (void)crt_set_option(gpu, "streams", 0);
        update(0.0f);
    }

    // tuning passthrough (crt.h "Options"), e.g. set_option("streams", 2) for multi-segment paths; false + `error` if refused
    bool set_option(const char* name, int value) {
        if (!gpu) return false;
        if (crt_set_option(gpu, name, value) != CRT_OK) { error = crt_last_error(); return false; }
        return true;
    }

    void Render() {                                  // Scene.h:1158-1231
        if (!gpu) return;
        if (camera.isMoving) {                       // Scene.h:1160-1172
            crt_reset(gpu);
            frame_count = 0;
            camera.isMoving = false;
        }
        const float r1 = rnd.randf2(), r2 = rnd.randf2();   // Scene.h:1208
        if (crt_render_frame(gpu, r1, r2) != CRT_OK) { error = crt_last_error(); return; }
        ++frame_count;                               // Scene.h:1223
    }

    // n frames of the loop above in one call: the same random vectors in the same order, the same sums bit for bit
    // (crt_render_frames shares launches among them on a one-segment path); for offline rendering, where nothing
    // happens between frames
    void RenderFrames(int n) {
        if (!gpu || n <= 0) return;
        if (camera.isMoving) { crt_reset(gpu); frame_count = 0; camera.isMoving = false; }
        std::vector<float> rx((size_t)n), ry((size_t)n);
        for (int i = 0; i < n; ++i) { rx[(size_t)i] = rnd.randf2(); ry[(size_t)i] = rnd.randf2(); }
        if (crt_render_frames(gpu, (uint32_t)n, rx.data(), ry.data()) != CRT_OK) { error = crt_last_error(); return; }
        frame_count += n;
    }

    void update(float /*second*/) {                  // Scene.h:1233-1246
        if (!gpu) return;
        const crt_camera c = camera.abi();
        crt_set_camera(gpu, &c);
    }

    // output pass (Shader/output.fs:9-20 with invSampleCounter = 1/frame_count, Scene.h:1224-1230); rows bottom-up
    std::vector<uint8_t> resolve() {
        std::vector<uint8_t> rgba((size_t)width * height * 4);
        if (gpu) crt_resolve(gpu, 1.0f / (float)(frame_count > 0 ? frame_count : 1), rgba.data(), rgba.size());
        return rgba;
    }
    std::vector<float> read_sum() {                  // path_trace_texture
        std::vector<float> rgb((size_t)width * height * 3);
        if (gpu) crt_read_sum(gpu, rgb.data(), rgb.size());
        return rgb;
    }
    // binary PPM, top row first (the buffers are bottom-up like GL)
    bool write_ppm(const std::string& path) {
        const std::vector<uint8_t> rgba = resolve();
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) return false;
        std::fprintf(f, "P6\n%u %u\n255\n", width, height);
        for (uint32_t y = 0; y < height; ++y) {
            const uint8_t* row = rgba.data() + (size_t)(height - 1 - y) * width * 4;
            for (uint32_t x = 0; x < width; ++x) std::fwrite(row + 4 * x, 1, 3, f);
        }
        std::fclose(f);
        return true;
    }

    // PNG (RGB), top row first
    bool write_png(const std::string& path) {
        const std::vector<uint8_t> rgba = resolve();
        std::vector<uint8_t> rgb((size_t)width * height * 3);
        for (size_t i = 0; i < (size_t)width * height; ++i) { rgb[3 * i] = rgba[4 * i]; rgb[3 * i + 1] = rgba[4 * i + 1]; rgb[3 * i + 2] = rgba[4 * i + 2]; }
        size_t size = 0;
        if (crt_image_encode_png(rgb.data(), (int32_t)width, (int32_t)height, 3, 1, nullptr, 0, &size) != CRT_OK) return false;
        std::vector<uint8_t> file(size);
        if (crt_image_encode_png(rgb.data(), (int32_t)width, (int32_t)height, 3, 1, file.data(), file.size(), &size) != CRT_OK) return false;
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) return false;
        const bool ok = std::fwrite(file.data(), 1, size, f) == size;
        std::fclose(f);
        return ok;
    }

    void delete_cpu_data() {                         // Scene.h:961-976
        std::vector<float3>().swap(vertices); std::vector<float3>().swap(normals); std::vector<float>().swap(texcoords);
        std::vector<crt_triangle>().swap(triangles); std::vector<crt_material>().swap(mats);
        std::vector<crt_light>().swap(lights); std::vector<crt_flatnode>().swap(flat_nodes);
        std::vector<uint8_t>().swap(albedo_textures_data);
    }
    void delete_gpu_data() {                         // Scene.h:978-998
        if (gpu) { crt_scene_destroy(gpu); gpu = nullptr; }
    }
};

}  // namespace crt
