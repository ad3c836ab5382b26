/*
 * ref_stb.c — builds the REFERENCE's own image decoder as a checker.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference loads `map_Kd` textures with `stbi_load(name, &w, &h, 0, 3)` (Caitlyn/Scene.h:619) from the stb_image it
 * vendors (Caitlyn/stb_image.h, v2.23).  That header is a self-contained C library, so — unlike the rest of the reference's
 * host code, which needs glm — it compiles here with gcc alone.  This translation unit only instantiates it FROM WHERE IT LIES
 * (-I/root/reference/Caitlyn; nothing of it is copied into this repository) and exports one entry point; the result goes to
 * oracle/_ref/libstbref.so (git-ignored).  It exists in the build container only: /root/reference is absent on the GPU box,
 * where the tests fall back to the committed fixtures this library produced (tests/golden/stb_decodes.npz,
 * tests/golden/make_stb_fixtures.py).
 */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_STDIO
#include "stb_image.h"

/* stbi_load_from_memory(..., 3): 8-bit RGB, top row first — what Scene.h:619 hands the texture path.  Returns 0 on failure;
 * the caller frees with ref_stbi_free. */
unsigned char* ref_stbi_load_rgb(const unsigned char* bytes, int len, int* w, int* h, int* channels_in_file) {
    return stbi_load_from_memory(bytes, len, w, h, channels_in_file, 3);
}
void ref_stbi_free(unsigned char* p) { stbi_image_free(p); }
const char* ref_stbi_failure(void) { return stbi_failure_reason(); }
