// Minimal host program with the reference's frame-loop shape (Caitlyn/main.cpp:244 init, :262-300 loop)
// on top of crt::Scene: load an OBJ, render N progressive frames on the GPU, write the tone-mapped image.
//
//   render_obj scene.obj out.ppm|out.png [width height frames max_depth [sum.f32]]
//
// Build: make -C caitlynrenderer_amd/csrc example   (g++, links libcrt.so)
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../caitlynrenderer_amd/csrc/host/scene.hpp"

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s scene.obj out.ppm|out.png [width height frames max_depth [sum.f32]]\n", argv[0]);
        return 2;
    }
    const uint32_t w = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 700, h = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 700;
    const int frames = argc > 5 ? std::atoi(argv[5]) : 16;
    const uint32_t depth = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 3;
    crt::Scene scn(argv[1], w, h, depth);                 // init_scene, main.cpp:28-67
    if (!scn.ok()) { std::fprintf(stderr, "scene failed: %s\n", scn.error.c_str()); return 1; }
    const bool batched = std::getenv("RENDER_OBJ_BATCHED") != nullptr;   // all frames through one crt_render_frames call
    if (const char* k = std::getenv("RENDER_OBJ_STREAMS"))               // tile shards of the frame side by side on k streams of the GPU
        if (!scn.set_option("streams", std::atoi(k))) { std::fprintf(stderr, "streams refused: %s\n", scn.error.c_str()); return 1; }
    scn.update(0.0f);
    if (batched) scn.RenderFrames(frames);
    else for (int i = 0; i < frames; ++i) {               // main.cpp:262-300: update(dt); render();
        scn.update(0.0f);
        scn.Render();
        if (!scn.error.empty()) { std::fprintf(stderr, "render failed: %s\n", scn.error.c_str()); return 1; }
    }
    if (!scn.error.empty()) { std::fprintf(stderr, "render failed: %s\n", scn.error.c_str()); return 1; }
    const std::string out = argv[2];
    const bool png = out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0;
    if (!(png ? scn.write_png(out) : scn.write_ppm(out))) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
    if (argc > 7) {
        const std::vector<float> sum = scn.read_sum();
        FILE* f = std::fopen(argv[7], "wb");
        if (!f) return 1;
        std::fwrite(sum.data(), sizeof(float), sum.size(), f);
        std::fclose(f);
    }
    std::printf("%d frames, %ux%u, depth %u -> %s\n", scn.frame_count, w, h, depth, argv[2]);
    return 0;
}
