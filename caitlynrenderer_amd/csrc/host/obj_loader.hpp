// Wavefront OBJ/MTL loader with the result semantics of the reference's
// Scene::Read_Object / ReadMtl (Caitlyn/Scene.h:186-315, :507-596, :742-926; SURVEY.md
// appendix B).  map_Kd textures (Scene.h:597-710) are decoded by host/image.cpp into the 256x256 RGB8 array the
// reference uploads (Scene.h:1065-1078).
#pragma once
#include <string>
#include <vector>

#include "../../../include/crt.h"
#include "vecmath.hpp"

namespace crt {

struct Mesh {
    std::vector<float3> vertices, normals;       // Scene.h:389-390
    std::vector<float> texcoords;                // uv pairs, v stored as 1-v (Scene.h:801)
    std::vector<crt_triangle> triangles;         // Scene.h:392, file order
    std::vector<crt_material> mats;              // Scene.h:399
    std::vector<crt_light> lights;               // Scene.h:400
    std::vector<uint8_t> albedo_textures;        // n_textures layers of tex_height x tex_width x RGB8 (Scene.h:408, :688-710)
    int tex_width = 256, tex_height = 256;       // Scene.h:57-58 require_tex_width / height
    int n_textures = 0;
    float3 vertex_min{1e20f, 1e20f, 1e20f};      // pre-translation minimum (Scene.h:767)
    float3 translation;                          // -vertex_min (Scene.h:917)
    std::string error;

    // Read_Object, Scene.h:742.  Translates vertices and light origins by -vertex_min
    // (Scene.h:915-925); the caller adds `translation` to the camera position.
    bool read_object(const std::string& file_name);
    bool read_mtl(const std::string& file_name, const std::string& directory, std::vector<std::pair<std::string, int>>& mtl_map);
};

}  // namespace crt
