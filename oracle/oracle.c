/*
 * oracle.c — scalar CPU restatement of CaitlynRenderer's per-ray hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (caitlynrenderer_amd/) never
 * calls, links or imports anything under oracle/.
 *
 * What it follows (all paths relative to /root/reference):
 *   rand                       Shader/path_trace.fs:38-42
 *   slab test                  Shader/path_trace.fs:84-109
 *   Moller-Trumbore            Shader/path_trace.fs:322-374 (record), :376-412 (no record)
 *   hit attributes             Shader/path_trace.fs:414-489
 *   BVH2 closest / any hit     Shader/path_trace.fs:511-667 / :669-819
 *   CWBVH node test + walks    Shader/cwbvh.fs:348-446, :448-536, :538-616 with the defects
 *                              listed in SURVEY.md §8a corrected as in SURVEY.md appendix C
 *   light sampling, integrator Shader/path_trace.fs:843-855, :857-1024
 *   ray generation, accumulate Shader/path_trace.fs:1026-1060
 *   resolve                    Shader/output.fs:9-20
 *   host RNG                   Caitlyn/Rnd.h:21-40
 *
 * How it is pinned.  The reference ships NO tests, golden images or fixtures for this path
 * (SURVEY.md §4), its GLSL cannot execute in this environment, and its only compilable host
 * header (sbvh.h) needs glm, which the image lacks, so no oracle/_ref build exists.  The
 * oracle is therefore pinned by the known-answer values the survey captured from the
 * reference's own sbvh.h and data files (committed under tests/golden/):
 * Cornell BVH2 leaf order and node boxes, host RNG sequence, camera after load, and the BVH2
 * primary-ray census (hit count, centre-pixel hit id and t, node/triangle visits per ray).
 * Bit-parity with a real GLSL run is UNPINNED: GLSL leaves sin/normalize/min/max precision
 * to the implementation, so the floating-point rules below are this project's definition.
 *
 * Floating-point rules (shared bit-for-bit with the HIP kernels):
 *   - compiled with -ffp-contract=off: no fused multiply-add except where fmaf() is written;
 *   - dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z; cross as usual; a*s then + left to right;
 *   - normalize(v) = v * (1.0f / sqrtf(dot(v,v))); length = sqrtf(dot); '/' and sqrtf are the
 *     IEEE correctly rounded operations;
 *   - min/max = fminf/fmaxf (IEEE minNum/maxNum: a NaN operand is ignored), which is also
 *     what v_min_f32/v_max_f32 do on gfx950;
 *   - sin/cos are evaluated in double by a fixed sequence of *, + and fused multiply-adds (orc_sin/orc_cos) and
 *     rounded once to float; tan(fov/2) is computed once per frame on the host with tanf.
 */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define ORC_INF 1e9f           /* path_trace.fs:35 */
#define ORC_EPS 1e-4f          /* path_trace.fs:36 */
#define ORC_PI 3.1415926f      /* path_trace.fs:16 */
#define ORC_PI2 6.2831853f     /* path_trace.fs:17 */
#define BVH2_STACK 128         /* reference: 12 / 16 ints (path_trace.fs:513, :671); too shallow at 2k tris */
#define BVH8_STACK 32          /* reference LOCAL_STACK_SIZE 16 (cwbvh.fs:374) */

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scl(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float len3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 norm3(v3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return scl(a, inv); }

/* ------------------------------------------------------------ pinned sin/cos -- */

/* Cody-Waite reduction by pi in three 33-bit pieces (the fdlibm pio2 constants doubled),
 * exact for |k| < 2^20, then a Taylor polynomial in double; error << 1 float ulp. */
static const double PI_1 = 0x1.921fb544p+1, PI_2 = 0x1.0b4611a6p-33, PI_3 = 0x1.3198a2ep-68;
static const double INV_PI = 0x1.45f306dc9c883p-2;

/* The sequence below is the definition shared with csrc/rt_math.hpp: Cody-Waite reduction by pi with three fused
 * multiply-adds, the two highest Taylor coefficients combined by a plain multiply and add, the rest of the Horner chain and the
 * final r + (r z) p / 1 - z p as fused multiply-adds (on the GPU: v_fma_f64 with the coefficient in an SGPR pair, 19 instead of
 * 32 half-rate instructions per call).  fma() is exactly defined by IEEE 754, so the clone that uses the FMA unit and the one
 * that calls libm's software fma give the same bits; target_clones picks at load time. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define ORC_FMA_CLONES __attribute__((target_clones("fma", "default")))
#else
#define ORC_FMA_CLONES
#endif
#define ORC_INLINE static inline __attribute__((always_inline))
ORC_INLINE double reduce_pi(double x, double* k_out) {
    double k = rint(x * INV_PI);
    double r = __builtin_fma(k, -PI_3, __builtin_fma(k, -PI_2, __builtin_fma(k, -PI_1, x)));
    *k_out = k;
    return r;
}
ORC_INLINE double sin_poly(double r) {
    double z = r * r;
    double p = z * (1.0 / 51090942171709440000.0) + (-1.0 / 121645100408832000.0);   /*  1/21!, -1/19! */
    p = __builtin_fma(p, z, 1.0 / 355687428096000.0);            /*  1/17! */
    p = __builtin_fma(p, z, -1.0 / 1307674368000.0);             /* -1/15! */
    p = __builtin_fma(p, z, 1.0 / 6227020800.0);                 /*  1/13! */
    p = __builtin_fma(p, z, -1.0 / 39916800.0);                  /* -1/11! */
    p = __builtin_fma(p, z, 1.0 / 362880.0);                     /*  1/9!  */
    p = __builtin_fma(p, z, -1.0 / 5040.0);                      /* -1/7!  */
    p = __builtin_fma(p, z, 1.0 / 120.0);                        /*  1/5!  */
    p = __builtin_fma(p, z, -1.0 / 6.0);                         /* -1/3!  */
    return __builtin_fma(r * z, p, r);
}
ORC_INLINE double cos_poly(double r) {
    double z = r * r;
    double p = z * (1.0 / 1124000727777607680000.0) + (-1.0 / 2432902008176640000.0);   /*  1/22!, -1/20! */
    p = __builtin_fma(p, z, 1.0 / 6402373705728000.0);           /*  1/18! */
    p = __builtin_fma(p, z, -1.0 / 20922789888000.0);            /* -1/16! */
    p = __builtin_fma(p, z, 1.0 / 87178291200.0);                /*  1/14! */
    p = __builtin_fma(p, z, -1.0 / 479001600.0);                 /* -1/12! */
    p = __builtin_fma(p, z, 1.0 / 3628800.0);                    /*  1/10! */
    p = __builtin_fma(p, z, -1.0 / 40320.0);                     /* -1/8!  */
    p = __builtin_fma(p, z, 1.0 / 720.0);                        /*  1/6!  */
    p = __builtin_fma(p, z, -1.0 / 24.0);                        /* -1/4!  */
    p = __builtin_fma(p, z, 0.5);                                /*  1/2!  */
    return __builtin_fma(-z, p, 1.0);
}
ORC_FMA_CLONES float orc_sin(float xf) {
    double x = (double)xf;
    if (!(fabs(x) < 1e9)) return 0.0f;                /* outside the domain the shaders reach */
    double k, r = reduce_pi(x, &k);
    double s = sin_poly(r);
    if (((long long)k) & 1) s = -s;
    return (float)s;
}
ORC_FMA_CLONES float orc_cos(float xf) {
    double x = (double)xf;
    if (!(fabs(x) < 1e9)) return 1.0f;
    double k, r = reduce_pi(x, &k);
    double c = cos_poly(r);
    if (((long long)k) & 1) c = -c;
    return (float)c;
}

/* path_trace.fs:38-42 */
float orc_rand(float seed[2], float rx, float ry) {
    float rv = rx * ry;
    seed[0] -= rv;
    seed[1] -= rv;
    float d = seed[0] * 12.9898f + seed[1] * 78.233f;
    float v = orc_sin(d) * 43758.5453f;
    return v - floorf(v);
}

/* Rnd.h:21-26 */
uint32_t orc_pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
/* Rnd.h:36-40, imax = 1.0f / UINT32_MAX (Rnd.h:8) */
float orc_randf2(uint32_t* state) {
    *state = orc_pcg_hash(*state);
    return (float)(*state) * (1.0f / (float)UINT32_MAX);
}

/* -------------------------------------------------------------- primitives -- */

typedef struct {
    float t, u, v;
    int slot;      /* index into s->triangles (BVH2 leaf order) */
    int id;        /* original triangle id */
    int mtl;
} rec_t;

typedef struct { uint32_t nodes, tris; } cnt_t;

static inline int orig_id(const orc_scene* s, int slot) { return s->tri_orig_ids ? s->tri_orig_ids[slot] : slot; }

/* path_trace.fs:322-374 (exact operation order). */
static inline int mt_test(const orc_scene* s, v3 o, v3 d, int slot, float* u, float* v, float* t) {
    const int32_t* ti = s->triangles + 12 * (size_t)slot;
    v3 v0 = ld3(s->vertices + 3 * (size_t)ti[0]);
    v3 v1 = ld3(s->vertices + 3 * (size_t)ti[1]);
    v3 v2 = ld3(s->vertices + 3 * (size_t)ti[2]);
    v1 = sub(v1, v0);
    v2 = sub(v2, v0);
    v3 pv = cross3(d, v2);
    v3 tv = sub(o, v0);
    v3 qv = cross3(tv, v1);
    float uu = dot3(tv, pv);
    float vv = dot3(d, qv);
    float tt = dot3(v2, qv);
    float inv_det = 1.0f / dot3(v1, pv);
    uu = uu * inv_det;
    vv = vv * inv_det;
    tt = tt * inv_det;
    float w = 1.0f - uu - vv;
    *u = uu; *v = vv; *t = tt;
    return (uu >= 0.0f) && (vv >= 0.0f) && (tt >= 0.0f) && (w >= 0.0f);
}

static inline void closest_update(const orc_scene* s, v3 o, v3 d, int slot, rec_t* rec, int tie, cnt_t* c) {
    float u, v, t;
    c->tris++;
    if (!mt_test(s, o, d, slot, &u, &v, &t)) return;
    int id = orig_id(s, slot);
    int better = t < rec->t;
    if (tie == ORC_TIE_LOWEST_ID && t == rec->t && rec->slot >= 0 && id < rec->id) better = 1;
    if (better) {
        rec->t = t; rec->u = u; rec->v = v; rec->slot = slot; rec->id = id;
        rec->mtl = s->triangles[12 * (size_t)slot + 3];
    }
}
/* path_trace.fs:376-412 */
static inline int any_test(const orc_scene* s, v3 o, v3 d, int slot, float max_t, cnt_t* c) {
    float u, v, t;
    c->tris++;
    return mt_test(s, o, d, slot, &u, &v, &t) && t < max_t;
}

/* path_trace.fs:84-109 */
static inline float hit_bbox(v3 o, v3 bmin, v3 bmax, v3 invdir, float* tl) {
    bmin = mul(sub(bmin, o), invdir);
    bmax = mul(sub(bmax, o), invdir);
    v3 tmax = V(fmaxf(bmax.x, bmin.x), fmaxf(bmax.y, bmin.y), fmaxf(bmax.z, bmin.z));
    v3 tmin = V(fminf(bmax.x, bmin.x), fminf(bmax.y, bmin.y), fminf(bmax.z, bmin.z));
    float th = fminf(tmax.x, fminf(tmax.y, tmax.z));
    *tl = fmaxf(tmin.x, fmaxf(tmin.y, tmin.z));
    return th;
}

/* ------------------------------------------------------------------- brute -- */

static void brute_closest(const orc_scene* s, v3 o, v3 d, float tmax, rec_t* rec, int tie, cnt_t* c) {
    rec->t = tmax; rec->slot = -1; rec->id = -1;
    for (int i = 0; i < s->n_triangles; ++i) closest_update(s, o, d, i, rec, tie, c);
}
static int brute_any(const orc_scene* s, v3 o, v3 d, float max_t, cnt_t* c) {
    for (int i = 0; i < s->n_triangles; ++i)
        if (any_test(s, o, d, i, max_t, c)) return 1;
    return 0;
}

/* -------------------------------------------------------------------- BVH2 -- */

/* path_trace.fs:511-652 */
static void bvh2_closest(const orc_scene* s, v3 o, v3 d, float tmax, rec_t* rec, int tie, cnt_t* c) {
    int stk[BVH2_STACK];
    int ptr = 0;
    stk[ptr++] = -1;
    rec->t = tmax; rec->slot = -1; rec->id = -1;
    const v3 invdir = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int ind = 0;
    while (ind > -1) {
        const float* nd = s->bvh2 + 8 * (size_t)ind;
        c->nodes++;
        int left = (int)nd[3];
        if (nd[7] == 0.0f) {
            int right = left + 1;
            const float* a = s->bvh2 + 8 * (size_t)left;
            const float* b = s->bvh2 + 8 * (size_t)right;
            float tl1, tl2;
            float th1 = hit_bbox(o, ld3(a), ld3(a + 4), invdir, &tl1);
            float th2 = hit_bbox(o, ld3(b), ld3(b + 4), invdir, &tl2);
            /* path_trace.fs:562-563 test `tl < t`; the lowest-id tie rule needs `<=` so that a
             * box holding an equal-t triangle is still entered */
            int l = th1 > 0 && th1 >= tl1 && (tie == ORC_TIE_LOWEST_ID ? tl1 <= rec->t : tl1 < rec->t);
            int r = th2 > 0 && th2 >= tl2 && (tie == ORC_TIE_LOWEST_ID ? tl2 <= rec->t : tl2 < rec->t);
            if (l) {
                ind = left;
                if (r) {
                    int off = tl1 > tl2 ? 1 : 0;
                    if (ptr < BVH2_STACK) stk[ptr++] = ind + 1 - off;
                    ind += off;
                }
                continue;
            } else if (r) {
                ind = right;
                continue;
            }
        } else {
            int range = (int)nd[7];
            for (int i = left; i < left + range; ++i) closest_update(s, o, d, i, rec, tie, c);
        }
        ind = stk[--ptr];
    }
}

/* path_trace.fs:669-819 */
static int bvh2_any(const orc_scene* s, v3 o, v3 d, float max_t, cnt_t* c) {
    int stk[BVH2_STACK];
    int ptr = 0;
    stk[ptr++] = -1;
    const v3 invdir = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int ind = 0;
    while (ind > -1) {
        const float* nd = s->bvh2 + 8 * (size_t)ind;
        c->nodes++;
        int left = (int)nd[3];
        if (nd[7] != 0.0f) {
            int range = (int)nd[7];
            for (int i = left; i < left + range; ++i)
                if (any_test(s, o, d, i, max_t, c)) return 1;
        } else {
            int right = left + 1;
            const float* a = s->bvh2 + 8 * (size_t)left;
            const float* b = s->bvh2 + 8 * (size_t)right;
            float tl1, tl2;
            float th1 = hit_bbox(o, ld3(a), ld3(a + 4), invdir, &tl1);
            float th2 = hit_bbox(o, ld3(b), ld3(b + 4), invdir, &tl2);
            int l = th1 >= 0 && th1 >= tl1 && tl1 <= max_t;
            int r = th2 >= 0 && th2 >= tl2 && tl2 <= max_t;
            if (l) {
                ind = left;
                if (r) {
                    int off = tl1 > tl2 ? 1 : 0;
                    if (ptr < BVH2_STACK) stk[ptr++] = ind + 1 - off;
                    ind += off;
                }
                continue;
            } else if (r) {
                ind = right;
                continue;
            }
        }
        ind = stk[--ptr];
    }
    return 0;
}

/* ------------------------------------------------------------------- CWBVH -- */

static inline uint32_t ld_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline float ld_f32(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int msb(uint32_t x) { return 31 - __builtin_clz(x); }   /* findMSB, x != 0 */

/* cwbvh.fs:348-353 */
static inline uint32_t oct_inv4(v3 d) {
    return (d.x < 0.0f ? 0u : 0x04040404u) | (d.y < 0.0f ? 0u : 0x02020202u) | (d.z < 0.0f ? 0u : 0x01010101u);
}
/* cwbvh.fs:369-372 */
static inline uint32_t sign_extend_s8x4(uint32_t x) { return ((x >> 7) & 0x01010101u) * 0xffu; }

/* cwbvh.fs:376-446 with: far plane = min(min()), tmin clamped to 0, tmax clamped to max_t,
 * child hit iff tmin <= tmax.  t = fmaf(q, 2^e*invdir, (p-o)*invdir). */
static uint32_t node8_intersect(const uint8_t* n, v3 o, int negx, int negy, int negz, v3 inv, uint32_t oct4, float max_t) {
    v3 p = V(ld_f32(n), ld_f32(n + 4), ld_f32(n + 8));
    uint32_t e_imask = ld_u32(n + 12);
    v3 adj_inv = V(u2f((e_imask & 0xffu) << 23) * inv.x, u2f(((e_imask >> 8) & 0xffu) << 23) * inv.y,
                   u2f(((e_imask >> 16) & 0xffu) << 23) * inv.z);
    v3 adj_o = mul(sub(p, o), inv);
    uint32_t hit_mask = 0;
    for (int i = 0; i < 2; ++i) {
        uint32_t meta4 = ld_u32(n + 24 + 4 * i);
        uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        uint32_t inner_mask4 = sign_extend_s8x4(is_inner4 << 3);
        uint32_t bit_index4 = (meta4 ^ (oct4 & inner_mask4)) & 0x1F1F1F1Fu;
        uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
        uint32_t qlox = ld_u32(n + 32 + 4 * i), qhix = ld_u32(n + 40 + 4 * i);
        uint32_t qloy = ld_u32(n + 48 + 4 * i), qhiy = ld_u32(n + 56 + 4 * i);
        uint32_t qloz = ld_u32(n + 64 + 4 * i), qhiz = ld_u32(n + 72 + 4 * i);
        uint32_t xmin = negx ? qhix : qlox, xmax = negx ? qlox : qhix;
        uint32_t ymin = negy ? qhiy : qloy, ymax = negy ? qloy : qhiy;
        uint32_t zmin = negz ? qhiz : qloz, zmax = negz ? qloz : qhiz;
        for (int j = 0; j < 4; ++j) {
            float tminx = fmaf((float)((xmin >> (8 * j)) & 0xffu), adj_inv.x, adj_o.x);
            float tminy = fmaf((float)((ymin >> (8 * j)) & 0xffu), adj_inv.y, adj_o.y);
            float tminz = fmaf((float)((zmin >> (8 * j)) & 0xffu), adj_inv.z, adj_o.z);
            float tmaxx = fmaf((float)((xmax >> (8 * j)) & 0xffu), adj_inv.x, adj_o.x);
            float tmaxy = fmaf((float)((ymax >> (8 * j)) & 0xffu), adj_inv.y, adj_o.y);
            float tmaxz = fmaf((float)((zmax >> (8 * j)) & 0xffu), adj_inv.z, adj_o.z);
            float tmin = fmaxf(fmaxf(tminx, tminy), fmaxf(tminz, 0.0f));
            float tmax = fminf(fminf(tmaxx, tmaxy), fminf(tmaxz, max_t));
            if (tmin <= tmax) {
                uint32_t child_bits = (child_bits4 >> (8 * j)) & 0xffu;
                uint32_t bit_index = (bit_index4 >> (8 * j)) & 0xffu;
                hit_mask |= child_bits << bit_index;
            }
        }
    }
    return hit_mask;
}

/* The shader forms 1/d directly (cwbvh.fs:460); with a zero direction component that makes
 * (p-o)*inf and q*inf+(-inf) NaN, the axis drops out of the slab test and the walk degenerates into
 * visiting everything the other axes overlap.  Such rays are common here: fract(sin()*43758.5453)
 * returns exactly 0 about once in 400 calls, which makes the cosine sample equal the (axis-aligned)
 * surface normal.  The usual CWBVH remedy is used: components smaller than 2^-80 in magnitude are
 * replaced by +-2^-80 for the traversal only (octant, reciprocal); Moller-Trumbore keeps the true d.
 * The slabs stay conservative, so hits are unchanged. */
static inline float clamp_dir(float d) {
    const float eps = 0x1p-80f;
    return fabsf(d) > eps ? d : copysignf(eps, d);
}

/* cwbvh.fs:448-536 (closest) and :538-616 (any): one walker, `any` returns at the first hit. */
static int bvh8_walk(const orc_scene* s, v3 o, v3 d, float tmax_in, int any, rec_t* rec, int tie, cnt_t* c) {
    uint32_t stack_x[BVH8_STACK], stack_y[BVH8_STACK];
    int sp = 0;
    float max_t = tmax_in;
    if (rec) { rec->t = tmax_in; rec->slot = -1; rec->id = -1; }
    /* a non-finite origin makes every slab NaN (all children pass): such a ray can hit nothing */
    if (!(isfinite(o.x) && isfinite(o.y) && isfinite(o.z))) return 0;
    const v3 dc = V(clamp_dir(d.x), clamp_dir(d.y), clamp_dir(d.z));
    const int negx = dc.x < 0.0f, negy = dc.y < 0.0f, negz = dc.z < 0.0f;
    const uint32_t oct4 = oct_inv4(dc);
    const v3 inv = V(1.0f / dc.x, 1.0f / dc.y, 1.0f / dc.z);
    uint32_t cur_x = 0, cur_y = 0x80000000u;
    for (;;) {
        uint32_t tri_x, tri_y;
        if (cur_y & 0xff000000u) {
            uint32_t hits_imask = cur_y;
            int off = msb(hits_imask);
            uint32_t base = cur_x;
            cur_y &= ~(1u << off);
            if (cur_y & 0xff000000u) {
                if (sp < BVH8_STACK) { stack_x[sp] = cur_x; stack_y[sp] = cur_y; sp++; }
            }
            uint32_t slot = (uint32_t)(off - 24) ^ (oct4 & 0xffu);
            uint32_t rel = (uint32_t)__builtin_popcount(hits_imask & ~(0xffffffffu << slot));
            uint32_t node_index = base + rel;
            const uint8_t* n = s->bvh8 + 80 * (size_t)node_index;
            c->nodes++;
            uint32_t hitmask = node8_intersect(n, o, negx, negy, negz, inv, oct4, max_t);
            uint32_t imask = n[15];
            cur_x = ld_u32(n + 16);
            tri_x = ld_u32(n + 20);
            cur_y = (hitmask & 0xff000000u) | imask;
            tri_y = hitmask & 0x00ffffffu;
        } else {
            tri_x = cur_x; tri_y = cur_y;
            cur_x = 0; cur_y = 0;
        }
        while (tri_y) {
            int b = msb(tri_y);
            tri_y &= ~(1u << b);
            int slot = s->bvh8_tri_slots[tri_x + (uint32_t)b];
            if (any) {
                if (any_test(s, o, d, slot, max_t, c)) return 1;
            } else {
                closest_update(s, o, d, slot, rec, tie, c);
                max_t = rec->t;
            }
        }
        if (!(cur_y & 0xff000000u)) {
            if (sp == 0) break;
            --sp;                                   /* cwbvh.fs:524 post-decrements; corrected */
            cur_x = stack_x[sp]; cur_y = stack_y[sp];
        }
    }
    return rec ? rec->slot >= 0 : 0;
}

/* --------------------------------------------------------- dispatch helpers -- */

static void closest(const orc_scene* s, int accel, int tie, v3 o, v3 d, float tmax, rec_t* rec, cnt_t* c) {
    if (accel == ORC_ACCEL_BVH8) bvh8_walk(s, o, d, tmax, 0, rec, tie, c);
    else if (accel == ORC_ACCEL_BVH2) bvh2_closest(s, o, d, tmax, rec, tie, c);
    else brute_closest(s, o, d, tmax, rec, tie, c);
}
static int occluded(const orc_scene* s, int accel, v3 o, v3 d, float max_t, cnt_t* c) {
    if (accel == ORC_ACCEL_BVH8) return bvh8_walk(s, o, d, max_t, 1, NULL, 0, c);
    if (accel == ORC_ACCEL_BVH2) return bvh2_any(s, o, d, max_t, c);
    return brute_any(s, o, d, max_t, c);
}

typedef struct {
    const orc_scene* s; int accel, mode, tie;
    const orc_ray* rays; orc_hit* hits; orc_ray_stats* stats;
    size_t i0, i1;
} trace_job;

static void* trace_worker(void* arg) {
    trace_job* j = (trace_job*)arg;
    for (size_t i = j->i0; i < j->i1; ++i) {
        const orc_ray* r = j->rays + i;
        v3 o = ld3(r->o), d = ld3(r->d);
        cnt_t c = {0, 0};
        orc_hit h = {0.f, 0.f, 0.f, -1};
        if (j->mode == ORC_ANY) {
            if (occluded(j->s, j->accel, o, d, r->tmax, &c)) h.tri = 0;
        } else {
            rec_t rec;
            closest(j->s, j->accel, j->tie, o, d, r->tmax, &rec, &c);
            if (rec.slot >= 0) { h.t = rec.t; h.u = rec.u; h.v = rec.v; h.tri = rec.id; }
        }
        j->hits[i] = h;
        if (j->stats) {
            j->stats[i].nodes = (uint16_t)(c.nodes > 65535 ? 65535 : c.nodes);
            j->stats[i].tris = (uint16_t)(c.tris > 65535 ? 65535 : c.tris);
        }
    }
    return NULL;
}

void orc_trace(const orc_scene* s, int accel, int mode, int tie, const orc_ray* rays, size_t n,
               orc_hit* hits, orc_ray_stats* stats, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    trace_job jobs[256];
    pthread_t th[256];
    size_t chunk = (n + (size_t)n_threads - 1) / (size_t)n_threads;
    for (int t = 0; t < n_threads; ++t) {
        size_t a = chunk * (size_t)t, b = a + chunk;
        if (a > n) a = n;
        if (b > n) b = n;
        trace_job j = {s, accel, mode, tie, rays, hits, stats, a, b};
        jobs[t] = j;
    }
    if (n_threads == 1) { trace_worker(&jobs[0]); return; }
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, trace_worker, &jobs[t]);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
}

/* ---------------------------------------------------------- ray generation -- */

/* path_trace.fs:1026-1047.  tex = (pixel centre) / resolution (Quad.h:16-24 + path_trace.vs). */
static void primary_ray(const orc_scene* s, int px, int py, float rx, float ry, int jitter, float seed[2],
                        v3* o, v3* dir) {
    const float W = (float)s->width, H = (float)s->height;
    seed[0] = (float)px + 0.5f;
    seed[1] = (float)py + 0.5f;
    float jx = 0.f, jy = 0.f;
    if (jitter) {
        float r1 = 2.0f * orc_rand(seed, rx, ry);
        float r2 = 2.0f * orc_rand(seed, rx, ry);
        jx = r1 < 1.0f ? sqrtf(r1) - 1.0f : 1.0f - sqrtf(2.0f - r1);
        jy = r2 < 1.0f ? sqrtf(r2) - 1.0f : 1.0f - sqrtf(2.0f - r2);
        jx = jx / (W * 0.5f);
        jy = jy / (H * 0.5f);
    }
    float tx = ((float)px + 0.5f) / W, ty = ((float)py + 0.5f) / H;
    float dx = (2.0f * tx - 1.0f) + jx;
    float dy = (2.0f * ty - 1.0f) + jy;
    float tan_fov = tanf(s->camera.fov * 0.5f);
    dx = dx * (W / H * tan_fov);
    dy = dy * tan_fov;
    v3 right = ld3(s->camera.right), up = ld3(s->camera.up), fwd = ld3(s->camera.forward);
    *dir = norm3(add(add(scl(right, dx), scl(up, dy)), fwd));
    *o = ld3(s->camera.position);
}

void orc_primary_rays(const orc_scene* s, float rx, float ry, int jitter, orc_ray* out) {
    for (int py = 0; py < s->height; ++py)
        for (int px = 0; px < s->width; ++px) {
            float seed[2];
            v3 o, d;
            primary_ray(s, px, py, rx, ry, jitter, seed, &o, &d);
            orc_ray* r = out + (size_t)py * (size_t)s->width + (size_t)px;
            r->o[0] = o.x; r->o[1] = o.y; r->o[2] = o.z; r->tmax = ORC_INF;
            r->d[0] = d.x; r->d[1] = d.y; r->d[2] = d.z; r->pad = 0;
        }
}

/* ---------------------------------------------------------------- integrator -- */

/* texture(albedo_textures, vec3(uv, layer)) as Scene.h:1065-1078 sets the array up: RGB8 UNORM, GL_LINEAR
 * min/mag filter, default GL_REPEAT wrap, no mipmaps.  GL leaves the filter arithmetic to the implementation;
 * this project's definition (shared with the kernel): texel = c / 255.0f; unnormalised coordinate x = u*W - 0.5;
 * i0 = floor(x), f = x - i0, indices wrapped with a mathematical modulo; top = t00*(1-fx) + t10*fx,
 * bot = t01*(1-fx) + t11*fx, result = top*(1-fy) + bot*fy. */
static inline int wrap_i(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
static v3 sample_albedo(const orc_scene* s, float u, float v, int layer) {
    const int W = s->tex_width, H = s->tex_height;
    if (layer < 0) layer = 0;
    if (layer > s->n_textures - 1) layer = s->n_textures - 1;
    const uint8_t* img = s->albedo_textures + (size_t)layer * (size_t)W * (size_t)H * 3;
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    /* huge or non-finite coordinates: keep the index arithmetic defined */
    if (!(fabsf(x0) < 1e9f)) { x0 = 0.f; fx = 0.f; }
    if (!(fabsf(y0) < 1e9f)) { y0 = 0.f; fy = 0.f; }
    int i0 = wrap_i((int)x0, W), i1 = wrap_i((int)x0 + 1, W);
    int j0 = wrap_i((int)y0, H), j1 = wrap_i((int)y0 + 1, H);
    float c[3];
    for (int k = 0; k < 3; ++k) {
        float t00 = (float)img[3 * ((size_t)j0 * W + i0) + k] / 255.0f, t10 = (float)img[3 * ((size_t)j0 * W + i1) + k] / 255.0f;
        float t01 = (float)img[3 * ((size_t)j1 * W + i0) + k] / 255.0f, t11 = (float)img[3 * ((size_t)j1 * W + i1) + k] / 255.0f;
        float top = t00 * (1.0f - fx) + t10 * fx;
        float bot = t01 * (1.0f - fx) + t11 * fx;
        c[k] = top * (1.0f - fy) + bot * fy;
    }
    return V(c[0], c[1], c[2]);
}

/* path_trace.fs:414-489 including the textured-albedo branch (:471-483). */
static void hit_attributes(const orc_scene* s, const rec_t* rec, v3* n, const float** mat, v3* albedo) {
    const int32_t* vn = s->triangles + 12 * (size_t)rec->slot + 4;
    if (vn[3] == 0) {
        *n = V((float)vn[0], (float)vn[1], (float)vn[2]);
    } else {
        v3 a = ld3(s->normals + 3 * (size_t)vn[0]);
        v3 b = ld3(s->normals + 3 * (size_t)vn[1]);
        v3 c = ld3(s->normals + 3 * (size_t)vn[2]);
        float w = 1.0f - rec->u - rec->v;      /* path_trace.fs:317-320 */
        *n = add(add(scl(a, w), scl(b, rec->u)), scl(c, rec->v));
    }
    *mat = s->materials + 16 * (size_t)rec->mtl;
    const float tex = (*mat)[12];
    if (tex != -1.0f && s->albedo_textures && s->n_textures > 0 && s->texcoords) {
        const int32_t* vt = s->triangles + 12 * (size_t)rec->slot + 8;
        const float* ta = s->texcoords + 2 * (size_t)vt[0];
        const float* tb = s->texcoords + 2 * (size_t)vt[1];
        const float* tc = s->texcoords + 2 * (size_t)vt[2];
        float w = 1.0f - rec->u - rec->v;                                   /* path_trace.fs:312-315 */
        float tu = (ta[0] * w + tb[0] * rec->u) + tc[0] * rec->v;
        float tv = (ta[1] * w + tb[1] * rec->u) + tc[1] * rec->v;
        v3 c = sample_albedo(s, tu, tv, (int)tex);
        /* pow(c, vec3(2.2f)): evaluated in double and rounded once (GLSL pow precision is implementation-defined) */
        *albedo = V((float)pow((double)c.x, (double)2.2f), (float)pow((double)c.y, (double)2.2f), (float)pow((double)c.z, (double)2.2f));
    } else {
        *albedo = ld3(*mat);
    }
}

/* path_trace.fs:214-218 */
static inline float power_heuristic(float a, float b) { float t = a * a; return t / (b * b + t); }

/* path_trace.fs:44-60 */
static inline void onb(v3 n, v3* bu, v3* bv) {
    if (n.z < -0.9999999f) { *bu = V(0.f, -1.f, 0.f); *bv = V(-1.f, 0.f, 0.f); }
    else {
        float a = 1.0f / (1.0f + n.z);
        float b = -n.x * n.y * a;
        *bu = V(1.0f + b, b, -n.x);
        *bv = V(b, 1.0f + b, -n.y);
    }
}

/* ------------------------------------------------ materials beyond Lambert --
 * NO REFERENCE CODE EXISTS for this block: the reference has the hooks only — the loader writes
 * albedo.w = Mirror_type for `type Mirror` (Scene.h:576-582), the enum names Mirror_type = 1 and Disney_type = 17
 * (Scene.h:111-132), the shader gates NEE on `specular.w == 0` (path_trace.fs:938) and tracks is_specular
 * (:865, :896, :1016), README.md:23 promises a "Disney BSDF" — and both shaders implement Lambert alone
 * (path_trace.fs:274-310).  What follows is THIS PROJECT'S definition (oracle-defined; parity with the reference is
 * unpinned by construction); the HIP kernel restates it operation for operation.
 *
 *   albedo.w == 1 (Mirror_type): perfect reflection d' = d - 2 (d.n) n about the normalised shading normal, throughput
 *       *= albedo.xyz, no NEE (a delta lobe cannot be light-sampled; this is what `specular.w` gates in the shader), no RNG
 *       draws, is_specular = true so that an emitter seen through the mirror takes the :896 branch.
 *   albedo.w == 17 (Disney_type): base colour = albedo.xyz (or the albedo texture), specular.x = metallic,
 *       specular.y = roughness.  Two lobes of Burley's 2012 model: the retro-reflective diffuse term weighted by
 *       (1 - metallic), and a GGX microfacet lobe (alpha = roughness^2, separable Smith G, Schlick Fresnel with
 *       F0 = lerp(0.04, base, metallic)).  One-sample MIS over the lobes: the specular lobe is picked with probability
 *       p = 0.5 + 0.5 metallic (GGX normal-distribution sampling), else cosine sampling; the pdf is the mixture.
 *       Unlike the reference's Lambert NEE (whose `diffuse_bsdf` returns the bare albedo, path_trace.fs:296-308), NEE here
 *       carries f * cos.  RNG draws per hit: 3 (NEE) + 3 (lobe pick, two for the direction). */
#define ORC_MIRROR_TYPE 1.0f
#define ORC_DISNEY_TYPE 17.0f

typedef struct { v3 base; float metallic, rough, a2, p_spec; } disney_t;

static inline float schlick5(float c) {
    float m = 1.0f - c;
    if (m < 0.0f) m = 0.0f;
    if (m > 1.0f) m = 1.0f;
    float m2 = m * m;
    return (m2 * m2) * m;
}
static inline float smith_g1(float c, float a2) { return (2.0f * c) / (c + sqrtf(a2 + (1.0f - a2) * (c * c))); }

static disney_t disney_params(v3 base, const float* specular) {
    disney_t m;
    m.base = base;
    float me = specular[0], ro = specular[1];
    m.metallic = me < 0.0f ? 0.0f : me > 1.0f ? 1.0f : me;
    m.rough = ro < 0.03f ? 0.03f : ro > 1.0f ? 1.0f : ro;      /* alpha >= 9e-4: D stays finite in fp32 */
    float a = m.rough * m.rough;
    m.a2 = a * a;
    m.p_spec = 0.5f + 0.5f * m.metallic;
    return m;
}

/* f(wo, wi) without the cosine, and the solid-angle pdf with which disney_sample picks wi; both 0 below the horizon.
 * n: unit shading normal on wo's side; wo = -ray direction. */
static void orc_disney_eval(const disney_t* m, v3 n, v3 wo, v3 wi, v3* f, float* pdf) {
    *f = V(0.f, 0.f, 0.f);
    *pdf = 0.f;
    float nl = dot3(n, wi), nv = dot3(n, wo);
    if (!(nl > 0.0f && nv > 0.0f)) return;
    v3 hs = add(wi, wo);
    float hh = dot3(hs, hs);
    if (!(hh > 0.0f)) return;
    v3 h = scl(hs, 1.0f / sqrtf(hh));
    float nh = dot3(n, h), lh = dot3(wi, h);
    if (!(nh > 0.0f && lh > 0.0f)) return;
    float fl = schlick5(nl), fv = schlick5(nv), fh = schlick5(lh);
    float fd90m1 = (0.5f + (2.0f * (lh * lh)) * m->rough) - 1.0f;
    float fd = (1.0f + fd90m1 * fl) * (1.0f + fd90m1 * fv);
    float kd = ((1.0f - m->metallic) * fd) / ORC_PI;
    float t = (nh * nh) * (m->a2 - 1.0f) + 1.0f;
    float D = m->a2 / ((ORC_PI * t) * t);
    float G = smith_g1(nl, m->a2) * smith_g1(nv, m->a2);
    float spec = (D * G) / ((4.0f * nl) * nv);
    float dm = 0.04f * (1.0f - m->metallic);
    v3 F0 = V(dm + m->base.x * m->metallic, dm + m->base.y * m->metallic, dm + m->base.z * m->metallic);
    v3 F = V(F0.x + (1.0f - F0.x) * fh, F0.y + (1.0f - F0.y) * fh, F0.z + (1.0f - F0.z) * fh);
    *f = add(scl(m->base, kd), scl(F, spec));
    float pdf_d = nl / ORC_PI;
    float pdf_s = (D * nh) / (4.0f * lh);
    *pdf = m->p_spec * pdf_s + (1.0f - m->p_spec) * pdf_d;
}

static v3 orc_disney_sample(const disney_t* m, v3 n, v3 wo, float u0, float u1, float u2) {
    v3 bu, bv;
    onb(n, &bu, &bv);
    float phi = ORC_PI2 * u2;
    if (u0 < m->p_spec) {                              /* GGX normal distribution -> half vector -> reflect wo */
        float c2 = (1.0f - u1) / (1.0f + (m->a2 - 1.0f) * u1);
        float s2 = 1.0f - c2;
        if (s2 < 0.0f) s2 = 0.0f;
        float ch = sqrtf(c2), sh = sqrtf(s2);
        v3 hl = V(sh * orc_cos(phi), sh * orc_sin(phi), ch);
        v3 h = add(add(scl(bu, hl.x), scl(bv, hl.y)), scl(n, hl.z));
        float vh = dot3(wo, h);
        return sub(scl(h, 2.0f * vh), wo);
    }
    float r = sqrtf(u1);                               /* path_trace.fs:257-270 */
    v3 dl = V(r * orc_cos(phi), r * orc_sin(phi), sqrtf(1.0f - u1));
    return add(add(scl(bu, dl.x), scl(bv, dl.y)), scl(n, dl.z));
}

/* test hooks (tests/test_materials.py: furnace / sampling-consistency checks of the lobe): params = base rgb, metallic, roughness */
void orc_disney_eval_test(const float params[5], const float n[3], const float wo[3], const float wi[3], float f[3], float* pdf) {
    float sp[2] = {params[3], params[4]};
    disney_t m = disney_params(ld3(params), sp);
    v3 fv;
    orc_disney_eval(&m, ld3(n), ld3(wo), ld3(wi), &fv, pdf);
    f[0] = fv.x; f[1] = fv.y; f[2] = fv.z;
}
void orc_disney_sample_test(const float params[5], const float n[3], const float wo[3], const float u[3], float wi[3]) {
    float sp[2] = {params[3], params[4]};
    disney_t m = disney_params(ld3(params), sp);
    v3 w = orc_disney_sample(&m, ld3(n), ld3(wo), u[0], u[1], u[2]);
    wi[0] = w.x; wi[1] = w.y; wi[2] = w.z;
}

/* path_trace.fs:857-1024; loop bound is max_depth (the shader hard-codes 3, :867).  The Mirror / Disney branches are the
 * oracle-defined extension above; a scene without such materials never enters them. */
static v3 path_trace(const orc_scene* s, int accel, int tie, v3 o, v3 d, float seed[2], float rx, float ry,
                     uint64_t counters[4]) {
    v3 L = V(0.f, 0.f, 0.f), T = V(1.f, 1.f, 1.f);
    float prev_pdf = 1.0f;
    int is_specular = 1;
    int true_area = 0;      /* the previous vertex was a Disney one: light pdfs use the triangle's true area (see below) */
    for (int i = 0; i < s->max_depth; ++i) {
        rec_t rec;
        cnt_t c = {0, 0};
        closest(s, accel, tie, o, d, ORC_INF, &rec, &c);
        counters[0]++; counters[2] += c.nodes; counters[3] += c.tris;
        if (rec.slot < 0) return L;
        v3 n, albedo; const float* mat;
        hit_attributes(s, &rec, &n, &mat, &albedo);
        float cos_incident = dot3(d, n);
        v3 original_n = n;
        if (cos_incident > 0) n = neg(n);
        const float* emission = mat + 4;
        if (emission[3] != -1.0f) {
            if (is_specular) return add(L, mul(T, ld3(emission)));
            v3 ld = scl(d, rec.t);
            float length = len3(ld);
            ld = norm3(ld);
            float cos_light = -1.0f * dot3(ld, n);
            float length2 = length * length;
            int li = (int)emission[3];
            const float* ap = s->lights + 18 * (size_t)li + 15;
            float pdf_light = length2 / (ap[0] * cos_light) * ap[1];
            if (true_area) pdf_light = 2.0f * pdf_light;
            float w = power_heuristic(prev_pdf, pdf_light);
            return add(L, scl(mul(T, ld3(emission)), w));
        }
        v3 hit_point = add(add(o, scl(d, rec.t)), scl(n, 0.0002f));
        const float* specular = mat + 8;
        const float type = mat[3];                                 /* albedo.w = MaterialType (Scene.h:111-132, :581) */
        if (type == ORC_MIRROR_TYPE) {
            v3 ns = norm3(n);
            float dn = dot3(d, ns);
            T = mul(T, albedo);
            is_specular = 1;
            o = hit_point;
            d = sub(d, scl(ns, 2.0f * dn));
            continue;
        }
        const int disney = type == ORC_DISNEY_TYPE;
        v3 ns = n, wo = neg(d);
        disney_t dm;
        if (disney) { ns = norm3(n); dm = disney_params(albedo, specular); }
        if (specular[3] == 0.0f && s->n_lights <= 0) {
            /* the shader would read light 0 of an empty buffer; keep the RNG stream, skip NEE */
            orc_rand(seed, rx, ry); orc_rand(seed, rx, ry); orc_rand(seed, rx, ry);
        } else if (specular[3] == 0.0f) {
            int li = (int)(orc_rand(seed, rx, ry) * (float)s->n_lights);
            if (li > s->n_lights - 1) li = s->n_lights - 1;   /* guards the one-ulp case rand*n == n */
            const float* Lt = s->lights + 18 * (size_t)li;
            float sq = sqrtf(orc_rand(seed, rx, ry));          /* path_trace.fs:843-855 */
            float b0 = 1.0f - sq;
            float b1 = orc_rand(seed, rx, ry) * sq;
            v3 lp = add(add(ld3(Lt), scl(ld3(Lt + 3), b0)), scl(ld3(Lt + 6), b1));
            v3 ldir = sub(lp, hit_point);
            float length = len3(ldir);
            float ilength = 1.0f / length;
            ldir = scl(ldir, ilength);
            float cos_mtl = dot3(ldir, original_n);
            float cos_light = dot3(ldir, ld3(Lt + 9));
            int lit = cos_mtl > 0.0f && cos_light < 0.0f;
            if (disney) lit = lit && dot3(ns, ldir) > 0.0f;        /* the lobe is zero below the shading horizon: no ray */
            if (lit) {
                cnt_t cs = {0, 0};
                int occ = occluded(s, accel, hit_point, ldir, length - ORC_EPS, &cs);
                counters[1]++; counters[2] += cs.nodes; counters[3] += cs.tris;
                if (!occ) {
                    v3 le = ld3(Lt + 12);
                    float pdf_light = (length * length) / (Lt[15] * -cos_light) * Lt[16];
                    v3 contrib;
                    if (disney) {
                        /* Light.area_pdf.x is |u x v| = TWICE the triangle's area (Scene.h:865-875) while the sample point is
                         * uniform on the triangle, so the reference's pdf_light is half the true density; the Lambert path keeps
                         * that (it is the reference's image), the Disney lobe uses the true one — here and in the MIS weight of
                         * the emitter hit that follows a Disney bounce — so that NEE + BSDF sampling add up (furnace test) */
                        pdf_light = 2.0f * pdf_light;
                        v3 f; float pdf_b;
                        orc_disney_eval(&dm, ns, wo, ldir, &f, &pdf_b);
                        float w = power_heuristic(pdf_light, pdf_b);
                        contrib = scl(mul(mul(T, le), f), dot3(ns, ldir) * w);
                    } else {
                        float bsdf_pdf = dot3(ldir, n) * 1.0f / ORC_PI;   /* `cos * ipi`, ipi = `1.0f / pi` unparenthesised (:18, :294) */
                        float w = power_heuristic(pdf_light, bsdf_pdf);
                        contrib = scl(mul(mul(T, le), albedo), w);
                    }
                    contrib = V(contrib.x / pdf_light, contrib.y / pdf_light, contrib.z / pdf_light);
                    L = add(L, contrib);
                }
            }
        }
        if (disney) {
            float u0 = orc_rand(seed, rx, ry);
            float u1 = orc_rand(seed, rx, ry);
            float u2 = orc_rand(seed, rx, ry);
            v3 wi = orc_disney_sample(&dm, ns, wo, u0, u1, u2);
            v3 f; float pdf;
            orc_disney_eval(&dm, ns, wo, wi, &f, &pdf);
            if (!(pdf > 0.0f)) return L;                           /* sampled below the horizon: the path ends */
            T = mul(T, scl(f, dot3(ns, wi) / pdf));
            prev_pdf = pdf;
            is_specular = 0;
            true_area = 1;
            o = hit_point;
            d = wi;
            continue;
        }
        /* path_trace.fs:44-60 onb, :257-270 cosine sample, :274-289 diffuse_sample */
        v3 bu, bv;
        onb(n, &bu, &bv);
        float u1 = orc_rand(seed, rx, ry);
        float u2 = orc_rand(seed, rx, ry);
        float r = sqrtf(u1);
        float phi = ORC_PI2 * u2;
        v3 dl = V(r * orc_cos(phi), r * orc_sin(phi), sqrtf(1.0f - u1));
        v3 sdir = add(add(scl(bu, dl.x), scl(bv, dl.y)), scl(n, dl.z));
        float bsdf_pdf = dot3(sdir, n) * 1.0f / ORC_PI;
        T = mul(T, albedo);
        prev_pdf = bsdf_pdf;
        is_specular = 0;
        true_area = 0;
        o = hit_point;
        d = sdir;
    }
    return L;
}

void orc_render_rows(const orc_scene* s, int accel, int tie, float rx, float ry, float* sum,
                     uint64_t counters[4], int y0, int y1) {
    for (int py = y0; py < y1; ++py)
        for (int px = 0; px < s->width; ++px) {
            float seed[2];
            v3 o, d;
            primary_ray(s, px, py, rx, ry, 1, seed, &o, &d);
            v3 c = path_trace(s, accel, tie, o, d, seed, rx, ry, counters);
            float* p = sum + 3 * ((size_t)py * (size_t)s->width + (size_t)px);
            p[0] = c.x + p[0]; p[1] = c.y + p[1]; p[2] = c.z + p[2];   /* path_trace.fs:1059 */
        }
}

typedef struct {
    const orc_scene* s; int accel, tie; float rx, ry; float* sum; uint64_t counters[4];
    volatile int* next_row; int rows_per_grab;
} frame_job;

static void* frame_worker(void* arg) {
    frame_job* j = (frame_job*)arg;
    for (;;) {
        int y0 = __sync_fetch_and_add(j->next_row, j->rows_per_grab);
        if (y0 >= j->s->height) break;
        int y1 = y0 + j->rows_per_grab;
        if (y1 > j->s->height) y1 = j->s->height;
        orc_render_rows(j->s, j->accel, j->tie, j->rx, j->ry, j->sum, j->counters, y0, y1);
    }
    return NULL;
}

void orc_render_frame(const orc_scene* s, int accel, int tie, float rx, float ry, float* sum,
                      uint64_t counters[4], int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    volatile int next_row = 0;
    frame_job jobs[256];
    pthread_t th[256];
    for (int t = 0; t < n_threads; ++t) {
        frame_job j = {s, accel, tie, rx, ry, sum, {0, 0, 0, 0}, &next_row, 4};
        jobs[t] = j;
    }
    if (n_threads == 1) frame_worker(&jobs[0]);
    else {
        for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, frame_worker, &jobs[t]);
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    }
    if (counters)
        for (int t = 0; t < n_threads; ++t)
            for (int k = 0; k < 4; ++k) counters[k] += jobs[t].counters[k];
}

/* Shader/output.fs:9-20: c = S*inv; c *= 1/(1 + lum/2); pow(c, 1/2.2); 8-bit UNORM write.
 * The power is PINNED (pow's last bits are the implementation's): the byte of x is the number of thresholds thr[1..255] it has reached,
 * thr[j] = the smallest float x >= 0 with (uint8)(clamp01((float)pow((double)x, 1/2.2)) * 255 + 0.5) >= j.  NaN and negatives reach none. */
static unsigned ref_gamma_byte(float x) {
    float v = (float)pow((double)x, (double)(1.0f / 2.2f));
    v = v < 0.f ? 0.f : v > 1.f ? 1.f : v;
    return (unsigned)(uint8_t)(v * 255.0f + 0.5f);
}
static const float* gamma_table(void) {
    static float thr[256];
    static int ready = 0;
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_mutex_lock(&mu);
    if (!ready) {
        thr[0] = 0.f;
        for (unsigned j = 1; j < 256; ++j) {
            uint32_t lo = 0u, hi = 0x7f800000u;            /* the non-negative floats by bit pattern */
            while (lo < hi) {
                uint32_t mid = lo + (hi - lo) / 2u;
                float x;
                memcpy(&x, &mid, 4);
                if (ref_gamma_byte(x) >= j) hi = mid; else lo = mid + 1u;
            }
            memcpy(&thr[j], &lo, 4);
        }
        ready = 1;
    }
    pthread_mutex_unlock(&mu);
    return thr;
}
void orc_resolve(const float* sum, size_t n_pixels, float inv_count, uint8_t* rgba) {
    const float* thr = gamma_table();
    for (size_t i = 0; i < n_pixels; ++i) {
        float c[3] = {sum[3 * i] * inv_count, sum[3 * i + 1] * inv_count, sum[3 * i + 2] * inv_count};
        float lum = 0.3f * c[0] + 0.6f * c[1] + 0.1f * c[2];
        float k = 1.0f / (1.0f + lum / 2.0f);
        for (int ch = 0; ch < 3; ++ch) {
            float x = c[ch] * 1.0f * k;
            /* number of thresholds reached = first j in [1, 256) with !(x >= thr[j]), minus one (thr ascends; a NaN reaches none) */
            unsigned lo = 1, hi = 256;
            while (lo < hi) {
                unsigned mid = (lo + hi) / 2;
                if (x >= thr[mid]) lo = mid + 1; else hi = mid;
            }
            rgba[4 * i + ch] = (uint8_t)(lo - 1);
        }
        rgba[4 * i + 3] = 255;
    }
}

int orc_hardware_threads(void) {
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n < 1 ? 1 : (int)n;
}
