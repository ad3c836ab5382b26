/*
 * oracle.h — CPU restatement of the reference's per-ray hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (caitlynrenderer_amd/) never does.  See oracle.c for the
 * reference file:line each function follows and for how the oracle is pinned.
 */
#ifndef ORACLE_H_
#define ORACLE_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ray { float o[3]; float tmax; float d[3]; uint32_t pad; } orc_ray;   /* == crt_ray */
typedef struct orc_hit { float t, u, v; int32_t tri; } orc_hit;                         /* == crt_hit */
typedef struct orc_ray_stats { uint16_t nodes, tris; } orc_ray_stats;

typedef struct orc_camera {
    float position[3], right[3], up[3], forward[3];
    float fov, focal_dist, aperture;
} orc_camera;

/* Borrowed views of the same buffers the reference uploads (Scene.h:1015-1062). */
typedef struct orc_scene {
    const float*   vertices;        /* xyz                                         */
    const float*   normals;         /* xyz                                         */
    const float*   texcoords;       /* uv (unused: no textures)                    */
    const int32_t* triangles;       /* 12 ints per triangle, BVH2 leaf order       */
    const int32_t* tri_orig_ids;    /* slot -> original id, may be NULL            */
    const float*   materials;       /* 16 floats per material; albedo.w = MaterialType (0 Lambert, 1 Mirror, 17 Disney), specular.xy = metallic, roughness */
    const float*   lights;          /* 18 floats per light                         */
    const float*   bvh2;            /* 8 floats per FlatNode                       */
    const uint8_t* bvh8;            /* 80 bytes per node8, may be NULL             */
    const int32_t* bvh8_tri_slots;  /* CWBVH triangle order -> slot in `triangles` */
    int32_t n_triangles, n_lights, n_bvh2, n_bvh8, n_bvh8_tris;
    int32_t width, height, max_depth;
    orc_camera camera;
    /* RGB8 albedo texture array (Scene.h:1065-1078): n_textures layers of tex_height x tex_width x 3; may be NULL */
    const uint8_t* albedo_textures;
    int32_t tex_width, tex_height, n_textures;
} orc_scene;

enum { ORC_CLOSEST = 0, ORC_ANY = 1 };
enum { ORC_TIE_FIRST_VISITED = 0,   /* path_trace.fs:363 strict '<' in traversal order   */
       ORC_TIE_LOWEST_ID = 1 };     /* t<best || (t==best && id<best_id), SURVEY app. C  */
enum { ORC_ACCEL_BRUTE = 0, ORC_ACCEL_BVH2 = 1, ORC_ACCEL_BVH8 = 2 };

/* pinned arithmetic */
float    orc_sin(float x);
float    orc_cos(float x);
float    orc_rand(float seed[2], float rx, float ry);          /* path_trace.fs:38-42 */
uint32_t orc_pcg_hash(uint32_t x);                             /* Rnd.h:21-26 */
float    orc_randf2(uint32_t* state);                          /* Rnd.h:36-40 */

/* explicit-ray traversal; stats may be NULL; hits[i].tri = original id or -1 */
void orc_trace(const orc_scene* s, int accel, int mode, int tie, const orc_ray* rays, size_t n,
               orc_hit* hits, orc_ray_stats* stats, int n_threads);

/* primary rays of one frame (path_trace.fs:1026-1047); jitter=0 gives pixel centres */
void orc_primary_rays(const orc_scene* s, float rx, float ry, int jitter, orc_ray* out);

/* one frame of the integrator added into sum[h][w][3] (bottom row first);
 * counters[0..3] += closest rays, any-hit rays, node fetches, triangle tests */
void orc_render_frame(const orc_scene* s, int accel, int tie, float rx, float ry, float* sum,
                      uint64_t counters[4], int n_threads);
/* same for pixel rows [y0,y1) only */
void orc_render_rows(const orc_scene* s, int accel, int tie, float rx, float ry, float* sum,
                     uint64_t counters[4], int y0, int y1);

/* Shader/output.fs:9-20 */
void orc_resolve(const float* sum, size_t n_pixels, float inv_count, uint8_t* rgba);

/* Test hooks of the oracle-defined Disney lobe (oracle.c "materials beyond Lambert"; no reference code exists for it):
 * params = base r, g, b, metallic, roughness; n = unit normal on wo's side.  eval: f(wo, wi) without the cosine and the
 * sampling pdf; sample: the direction disney_sample picks for the uniform triple u. */
void orc_disney_eval_test(const float params[5], const float n[3], const float wo[3], const float wi[3], float f[3], float* pdf);
void orc_disney_sample_test(const float params[5], const float n[3], const float wo[3], const float u[3], float wi[3]);

int orc_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
