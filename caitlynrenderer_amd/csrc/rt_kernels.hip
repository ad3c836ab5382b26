// HIP kernels for gfx950 (CDNA4): CWBVH traversal (closest / any hit) and the thin wavefront
// shell around it (ray generation, shading + NEE ray emission, shadow resolve, accumulate).
//
// Reference behaviour restated here (never its code): Shader/cwbvh.fs:348-616 (traversal, with the
// SURVEY.md §8a defects corrected), Shader/path_trace.fs:322-374 (Moller-Trumbore), :414-489 (hit
// attributes), :843-1024 (integrator), :1026-1060 (ray generation, accumulate), Shader/output.fs
// (resolve).  One ray per lane, 64-lane wavefronts; traversal stack in LDS; queue compaction with
// wave ballots.  No MFMA: this is pointer chasing; measured bound = VALU issue (primary / shadow rays) or L2-miss
// latency (bounce rays), see DESIGN.md section 5.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "rt_kernels.hpp"
#include "rt_math.hpp"
#include "host/flatnode_link.hpp"

namespace crt {

#define CRT_INF 1e9f        // path_trace.fs:35
#define CRT_EPS 1e-4f       // path_trace.fs:36
#define CRT_PI 3.1415926f   // path_trace.fs:16
#define CRT_PI2 6.2831853f  // path_trace.fs:17

// ------------------------------------------------------------------ traversal --------

// `make asm` builds with -DCRT_ISA_MARKS: comment lines in the assembly bracket the traversal loop and its node / triangle steps so
// that tools/roofline.py can count the vector instructions one node visit and one triangle test cost (the roofline's
// algorithmic instruction count).  The product build carries no markers: an inline-asm statement is a scheduling barrier.
#ifdef CRT_ISA_MARKS
#define CRT_MARK(name) asm volatile("; CRT_MARK " name)
#else
#define CRT_MARK(name) ((void)0)
#endif

__device__ __forceinline__ uint32_t sign_extend_s8x4(uint32_t x) { return ((x >> 7) & 0x01010101u) * 0xffu; }   // cwbvh.fs:369-372


__device__ __forceinline__ float ubyte_f(uint32_t x, int j) { return (float)((x >> (8 * j)) & 0xffu); }          // v_cvt_f32_ubyteN

// 8-wide quantised child-box test (cwbvh.fs:376-446, corrected: far = min(min()), tmin clamped to 0,
// tmax clamped to max_t, hit iff tmin <= tmax).  ~19 VALU instructions per child (6 cvt_f32_ubyte, 6 fma, max3,
// min3, 2 clamps, compare, shift, select); pairing the near/far fmas of an axis into v_pk_fma_f32 (24 instead of
// 48) was measured twice: 0.224 vs 0.217 ms (200-frame averages) and 9 more VGPRs, so the scalar form stays.  Returns the hit mask: inner children in the top
// byte at bit (24+slot)^oct, leaf triangles as unary-count bits in the low 24.
// Round 3 measured what each instruction kind costs (profiles/r03_valu_issue_cycles.txt: only fma / mul / add / mov issue at ~2.5
// cycles per wave64 instruction, conversions, min / max, compares and integer ops at ~4.2) and tried the obvious answer — a 128-byte
// device copy of the node with the planes widened to IEEE halves, so that the conversion rides inside v_fma_mix_f32 and the ray picks
// near / far plane rows by address instead of 12 v_cndmask (55 fewer instructions per node, same arithmetic, bit-identical).  It
// lost: v_fma_mix_f32 is a 4.3-cycle instruction itself, and 8 row loads per node instead of 5 saturate the CU's vector-memory path
// (1,004,672 triangles 12,574 vs 12,365 Mray/s at 96 VGPRs but the 80-VGPR build spills in the loop, 4 segments 4,995 vs 5,047,
// the 8 M-triangle scene 6,942 vs 8,588).  The patch is kept as profiles/r03_f16_planes_experiment.patch.
// Also tried: the mask assembly (54 of the 230 instructions, all of the 4.2-cycle kind) with the byte extractions folded into SDWA
// operand selects by inline assembly (v_lshlrev_b32_sdwa: child_bits << bit_index in one instruction per child, the exponent bytes
// likewise): 223 instructions per visit instead of 230 and SLOWER — 13,120 vs 13,660 Mray/s, 4 segments 5,357 vs 5,402, Cornell
// 45,250 vs 44,640: an SDWA instruction costs more issue time than the two plain ones it replaces.
__device__ __forceinline__ uint32_t node8_intersect(const uint4 n0, const uint4 n1, const uint4 n2, const uint4 n3,
                                                    const uint4 n4, vec3 o, vec3 inv, bool negx, bool negy, bool negz,
                                                    uint32_t oct4, float max_t) {
    const vec3 p = V3(__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z));
    const uint32_t e_imask = n0.w;
    const vec3 adj_inv = V3(__uint_as_float((e_imask & 0xffu) << 23) * inv.x,
                            __uint_as_float(((e_imask >> 8) & 0xffu) << 23) * inv.y,
                            __uint_as_float(((e_imask >> 16) & 0xffu) << 23) * inv.z);
    const vec3 adj_o = (p - o) * inv;
    uint32_t hit_mask = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t meta4 = i == 0 ? n1.z : n1.w;
        const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t inner_mask4 = sign_extend_s8x4(is_inner4 << 3);
        const uint32_t bit_index4 = (meta4 ^ (oct4 & inner_mask4)) & 0x1F1F1F1Fu;
        const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
        const uint32_t qlox = i == 0 ? n2.x : n2.y, qhix = i == 0 ? n2.z : n2.w;
        const uint32_t qloy = i == 0 ? n3.x : n3.y, qhiy = i == 0 ? n3.z : n3.w;
        const uint32_t qloz = i == 0 ? n4.x : n4.y, qhiz = i == 0 ? n4.z : n4.w;
        const uint32_t xmin = negx ? qhix : qlox, xmax = negx ? qlox : qhix;
        const uint32_t ymin = negy ? qhiy : qloy, ymax = negy ? qloy : qhiy;
        const uint32_t zmin = negz ? qhiz : qloz, zmax = negz ? qloz : qhiz;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float tminx = __builtin_fmaf(ubyte_f(xmin, j), adj_inv.x, adj_o.x);
            const float tminy = __builtin_fmaf(ubyte_f(ymin, j), adj_inv.y, adj_o.y);
            const float tminz = __builtin_fmaf(ubyte_f(zmin, j), adj_inv.z, adj_o.z);
            const float tmaxx = __builtin_fmaf(ubyte_f(xmax, j), adj_inv.x, adj_o.x);
            const float tmaxy = __builtin_fmaf(ubyte_f(ymax, j), adj_inv.y, adj_o.y);
            const float tmaxz = __builtin_fmaf(ubyte_f(zmax, j), adj_inv.z, adj_o.z);
            const float tmin = __builtin_fmaxf(__builtin_fmaxf(tminx, tminy), __builtin_fmaxf(tminz, 0.0f));
            const float tmax = __builtin_fminf(__builtin_fminf(tmaxx, tmaxy), __builtin_fminf(tmaxz, max_t));
            if (tmin <= tmax) {
                const uint32_t child_bits = (child_bits4 >> (8 * j)) & 0xffu;
                const uint32_t bit_index = (bit_index4 >> (8 * j)) & 0xffu;
                hit_mask |= child_bits << bit_index;
            }
        }
    }
    return hit_mask;
}

__device__ __forceinline__ float clamp_dir(float d) {
    const float eps = 0x1p-80f;
    return __builtin_fabsf(d) > eps ? d : __builtin_copysignf(eps, d);
}

// counting kernels only: +1 on exactly one lane each time the wave executes the enclosing block (wave-level step counter;
// lane visits / (64 x wave steps) = the lane utilisation of that block)
__device__ __forceinline__ void count_wave_step(uint32_t& c) {
    const unsigned long long m = __ballot(true);
    if ((int)(threadIdx.x & 63u) == __builtin_ctzll(m)) ++c;
}

// Measurement aid (crt_debug_step_hist): how many lanes were enabled at each node step of the counting kernels, per walk kind — the
// distribution behind the lane-utilisation figures (how much of the idle time is "fewer than half of the lanes still have a ray").
__device__ unsigned long long* g_step_hist = nullptr;       // [2][65]: closest-hit walks, any-hit walks; null = off
__device__ uint32_t g_step_hist_mode = 0u;                 // 0: by enabled lanes; 1: by DISTINCT (node, octant) keys among the enabled lanes (1 = a uniform step)
__device__ __forceinline__ void hist_node_step(bool any, uint32_t nidx = 0u, uint32_t oct = 0u) {
    unsigned long long* const h = g_step_hist;
    if (h == nullptr) return;
    const unsigned long long m = __ballot(true);
    uint32_t n = (uint32_t)__builtin_popcountll(m);
    if (g_step_hist_mode) {                                 // what a packet walk (one scalar-unit step per distinct node of the wave) would have to execute
        const uint32_t key = (nidx << 3) | (oct & 7u);
        unsigned long long left = m;
        n = 0u;
        while (left) {
            const uint32_t k0 = (uint32_t)__shfl((int)key, __builtin_ctzll(left));
            left &= ~__ballot(key == k0);
            ++n;
        }
    }
    if ((int)(threadIdx.x & 63u) == __builtin_ctzll(m)) atomicAdd(&h[(any ? 65 : 0) + n], 1ull);
}
void set_step_hist(unsigned long long* d_hist, uint32_t mode) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_step_hist), &d_hist, sizeof d_hist);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_step_hist_mode), &mode, sizeof mode);
}

struct HitState {
    float t, u, v;
    int tri;   // index into the CWBVH-ordered triangle array, -1 = none
    int id;    // original triangle id of `tri`
};

// Moller-Trumbore, operation order of path_trace.fs:337-360, on the pre-gathered record
// (v0, e1 = v1 - v0, e2 = v2 - v0): the two subtractions are the same fp32 operations the shader
// performs per test, done once at upload.
__device__ __forceinline__ bool mt_test(const float4 a, const float4 b, const float4 c, vec3 o, vec3 d, float& u,
                                        float& v, float& t) {
    const vec3 v0 = V3(a.x, a.y, a.z), e1 = V3(b.x, b.y, b.z), e2 = V3(c.x, c.y, c.z);
    const vec3 pv = cross(d, e2);
    const vec3 tv = o - v0;
    const vec3 qv = cross(tv, e1);
    float uu = dot(tv, pv);
    float vv = dot(d, qv);
    float tt = dot(e2, qv);
    const float inv_det = rcp_ieee(dot(e1, pv));
    uu = uu * inv_det;
    vv = vv * inv_det;
    tt = tt * inv_det;
    const float w = 1.0f - uu - vv;
    u = uu; v = vv; t = tt;
    return (uu >= 0.0f) & (vv >= 0.0f) & (tt >= 0.0f) & (w >= 0.0f);
}

// Row address of a node / triangle record as uniform base + a 32-bit byte offset: the load then takes the base from an SGPR pair and the
// offset from one VGPR (global_load ... v_off, s[base]) instead of a 64-bit v_mad_u64_u32 per visit — an instruction that costs 8.9
// issue cycles against 4.4 for the 32-bit multiply (profiles/r03_valu_issue_cycles.txt).  crt_scene_create refuses arrays beyond 4 GiB.
__device__ __forceinline__ const uint4* node_rows(const uint4* nodes, uint32_t idx) {
    return reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(nodes) + (size_t)(idx * (uint32_t)(CRT_NODE_ROWS * 16)));
}
__device__ __forceinline__ const float4* tri_rows(const float4* tris, uint32_t idx) {
    return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tris) + (size_t)(idx * (uint32_t)(CRT_TRI_ROWS * 16)));
}

// ---- UNIFORM NODE STEPS (round 4): the node through the scalar cache ----
// The 64 primary rays of a wave leave a 4 x 4 pixel quadrant, so near the root they all ask for the SAME node and share the direction
// octant: 45 % of the first segment's closest-hit node steps and 30 % of its shadow walks' (8 x 8 pixel waves; profiles/r04_experiments.md).
// Such a step needs no vector load at all: one lane's index, five s_load_dwordx4 through the scalar data cache, and the node sits in 20
// SGPRs.  Everything that depends on the node and the octant alone — the three exponents, `meta` (inner mask, bit index, child bits), the
// near / far plane selects, child_bits << bit_index of all eight children — then runs on the scalar unit; what is left per lane is the
// arithmetic of node8_intersect itself on the same operands (conversions read the bytes from SGPRs), so the hit mask keeps its bits.
// uniform_node_key: what must agree across the enabled lanes (node index, octant).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool node_step_is_uniform(uint32_t nidx, uint32_t oct4, uint32_t& key0) {
    const uint32_t key = (nidx << 3) | (oct4 & 7u);          // crt_scene_create keeps node offsets below 4 GiB: nidx < 2^26
    key0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
    // "no lane disagrees" must not hold vacuously: scalar loads ignore EXEC, so a block reached with no lane enabled would read the node of a
    // stale key (the compiler skips such blocks today; nothing guarantees it)
    return __ballot(true) != 0ull && __ballot(key != key0) == 0ull;
}
__device__ __forceinline__ void load_node_scalar(const uint4* nodes, uint32_t nidx0, uint4& n0, uint4& n1, uint4& n2, uint4& n3, uint4& n4) {
    const char* p = reinterpret_cast<const char*>(nodes) + (size_t)nidx0 * (size_t)(CRT_NODE_ROWS * 16);
    u32x4 a, b, c, d, e;
    // inline assembly: the compiler would pick a vector load here (it cannot prove that none of the kernel's stores touches the node array)
    asm volatile("s_load_dwordx4 %0, %5, 0x0\n\ts_load_dwordx4 %1, %5, 0x10\n\ts_load_dwordx4 %2, %5, 0x20\n\ts_load_dwordx4 %3, %5, 0x30\n\t"
                 "s_load_dwordx4 %4, %5, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d), "=&s"(e) : "s"(p));
    n0 = make_uint4(a.x, a.y, a.z, a.w); n1 = make_uint4(b.x, b.y, b.z, b.w); n2 = make_uint4(c.x, c.y, c.z, c.w);
    n3 = make_uint4(d.x, d.y, d.z, d.w); n4 = make_uint4(e.x, e.y, e.z, e.w);
}
// The uniform step without the 48 byte-to-float conversions: the scene keeps every node's child planes as floats as well (12 rows of
// float4 per node, SegmentArgs::planes; (float)byte is exact), the step fetches the rows it needs through the scalar cache — near / far
// row of each axis picked by ADDRESS from the shared octant — and each fma reads its plane straight from an SGPR.  Same operands, same
// fma: the hit mask keeps its bits.  Two halves of four children (24 SGPRs of planes at a time).
__device__ __forceinline__ uint32_t node8_intersect_planes(uint4& n0, uint4& n1, const uint4* nodes, const float4* planes, uint32_t nidx0, vec3 o, vec3 inv,
                                                           uint32_t oct0, float max_t) {
    // byte offsets of the near / far rows inside a half: axis * 32 + side * 16, side = hi planes when the direction is negative (oct bit clear)
    const uint32_t sx = (oct0 & 4u) ? 0u : 16u, sy = (oct0 & 2u) ? 0u : 16u, sz = (oct0 & 1u) ? 0u : 16u;
    const char* base = reinterpret_cast<const char*>(planes) + (size_t)nidx0 * 192u;
    const char* np = reinterpret_cast<const char*>(nodes) + (size_t)nidx0 * (size_t)(CRT_NODE_ROWS * 16);
    u32x4 row[2][6];
    u32x4 a, b;
    asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(np));
    n0 = make_uint4(a.x, a.y, a.z, a.w); n1 = make_uint4(b.x, b.y, b.z, b.w);
    const vec3 p = V3(__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z));
    const uint32_t e_imask = n0.w;
    const vec3 adj_inv = V3(__uint_as_float((e_imask & 0xffu) << 23) * inv.x, __uint_as_float(((e_imask >> 8) & 0xffu) << 23) * inv.y,
                            __uint_as_float(((e_imask >> 16) & 0xffu) << 23) * inv.z);
    const vec3 adj_o = (p - o) * inv;
    const uint32_t oct4 = oct0 * 0x01010101u;
    uint32_t hit_mask = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        {
        const char* bh = base + h * 96;
        asm volatile("s_load_dwordx4 %0, %6, %7\n\ts_load_dwordx4 %1, %6, %8\n\ts_load_dwordx4 %2, %6, %9\n\ts_load_dwordx4 %3, %6, %10\n\t"
                     "s_load_dwordx4 %4, %6, %11\n\ts_load_dwordx4 %5, %6, %12\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(row[h][0]), "=&s"(row[h][1]), "=&s"(row[h][2]), "=&s"(row[h][3]), "=&s"(row[h][4]), "=&s"(row[h][5])
                     : "s"(bh), "s"(sx), "s"(16u - sx), "s"(32u + sy), "s"(48u - sy), "s"(64u + sz), "s"(80u - sz));
        }
        const u32x4 xn = row[h][0], xf = row[h][1], yn = row[h][2], yf = row[h][3], zn = row[h][4], zf = row[h][5];
        const uint32_t meta4 = h == 0 ? n1.z : n1.w;
        const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t inner_mask4 = sign_extend_s8x4(is_inner4 << 3);
        const uint32_t bit_index4 = (meta4 ^ (oct4 & inner_mask4)) & 0x1F1F1F1Fu;
        const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float tminx = __builtin_fmaf(__uint_as_float(xn[j]), adj_inv.x, adj_o.x);
            const float tminy = __builtin_fmaf(__uint_as_float(yn[j]), adj_inv.y, adj_o.y);
            const float tminz = __builtin_fmaf(__uint_as_float(zn[j]), adj_inv.z, adj_o.z);
            const float tmaxx = __builtin_fmaf(__uint_as_float(xf[j]), adj_inv.x, adj_o.x);
            const float tmaxy = __builtin_fmaf(__uint_as_float(yf[j]), adj_inv.y, adj_o.y);
            const float tmaxz = __builtin_fmaf(__uint_as_float(zf[j]), adj_inv.z, adj_o.z);
            const float tmin = __builtin_fmaxf(__builtin_fmaxf(tminx, tminy), __builtin_fmaxf(tminz, 0.0f));
            const float tmax = __builtin_fminf(__builtin_fminf(tmaxx, tmaxy), __builtin_fminf(tmaxz, max_t));
            if (tmin <= tmax) {
                const uint32_t child_bits = (child_bits4 >> (8 * j)) & 0xffu;
                const uint32_t bit_index = (bit_index4 >> (8 * j)) & 0xffu;
                hit_mask |= child_bits << bit_index;
            }
        }
    }
    return hit_mask;
}
// Build switches that remain (make EXTRA="-D<name>=<value>"; tools/variant.sh builds a variant library beside the product one):
//   CRT_SEG_OCC, CRT_SEG_OCC_FIRST, CRT_SEG_OCC_BATCH, CRT_SEG_OCC_DEFERRED, CRT_SHADOW_OCC   waves per SIMD the segment kernels / k_shadow_deferred are compiled for (6 / 6 / 5 / 8 / 8)
//   CRT_HIT_SLOTS, CRT_NODE_ROWS, CRT_TRI_ROWS           LDS hit-record slots per lane; device row strides (rt_kernels.hpp)
//   CRT_ISA_MARKS                                        `make asm`: marker comments tools/roofline.py counts between
//   CRT_EXPERIMENTS                                      persistent grids and 2- / 4-wave workgroups (make EXPERIMENTS=1)
// Every variant that lost its measurement twice (flag-carrying loops, node touches, LDS-staged group nodes, one-wait plane loads, shared
// triangle steps, shadow-ray compaction across waves, the lean single / bounce builds ...) is in the git history and in
// profiles/r04_experiments.md, not here.
constexpr uint32_t GROUP_KL = 3;     // lanes per ray after the regroup = 1 << this: 8 (one regroup, when at most 8 rays are left: 6,021 Mray/s on four
                                     // segments of the 1 M-triangle scene against 5,464 with quads at <= 16 and 5,438 with pairs at <= 32)

// One ray through the CWBVH (cwbvh.fs:448-536 closest, :538-616 any).  `stk` is this lane's column
// of the wave's LDS stack: stk[level * 64]; stack_entries (<= CRT_STACK_ENTRIES) is sized from the
// CWBVH's depth at scene creation so shallow trees leave more LDS for occupancy.
template <bool ANY, bool STATS, bool UNI = false>
__device__ __forceinline__ bool traverse(const uint4* __restrict__ nodes, const float4* __restrict__ tris, vec3 o,
                                         vec3 d, float tmax_in, uint2* stk, int stack_entries, uint32_t* overflow, HitState& best,
                                         uint32_t& n_nodes, uint32_t& n_tris, uint32_t& w_nodes, uint32_t& w_tris, uint32_t* n_uni = nullptr) {
    best.t = tmax_in; best.u = 0.f; best.v = 0.f; best.tri = -1; best.id = -1;
    // a non-finite origin makes every slab NaN (all children pass): such a ray can hit nothing
    if (!(__builtin_isfinite(o.x) && __builtin_isfinite(o.y) && __builtin_isfinite(o.z))) return false;
    // Zero direction components (common: the shader RNG returns exactly 0 once in ~400 calls, which makes
    // the cosine sample equal an axis-aligned normal) would turn (p-o)*inf into NaN and drop the axis from
    // the slab test.  Traversal uses +-2^-80 instead (octant and reciprocal only); the triangle test keeps d.
    const vec3 dc = V3(clamp_dir(d.x), clamp_dir(d.y), clamp_dir(d.z));
    const bool negx = dc.x < 0.0f, negy = dc.y < 0.0f, negz = dc.z < 0.0f;
    const uint32_t oct4 = (negx ? 0u : 0x04040404u) | (negy ? 0u : 0x02020202u) | (negz ? 0u : 0x01010101u);   // cwbvh.fs:348-353
    const vec3 inv = V3(rcp_ieee(dc.x), rcp_ieee(dc.y), rcp_ieee(dc.z));
    float max_t = tmax_in;
    int sp = 0;
    uint2 cur = make_uint2(0u, 0x80000000u);
    CRT_MARK("loop_begin plain");
    for (;;) {
        uint2 tg;
        if (cur.y & 0xff000000u) {
            CRT_MARK("node_begin");
            const uint32_t hits_imask = cur.y;
            const int off = 31 - __builtin_clz(hits_imask);
            const uint32_t base = cur.x;
            cur.y &= ~(1u << off);
            if (cur.y & 0xff000000u) {
                // crt_scene_create sizes the stack from the validated depth of the tree, so a push always fits; if a
                // caller-supplied tree ever got past the validator the dropped push is counted, not silent
                if (sp < stack_entries) { stk[sp * 64] = cur; ++sp; } else atomicAdd(overflow, 1u);
            }
            const uint32_t slot = (uint32_t)(off - 24) ^ (oct4 & 0xffu);
            const uint32_t rel = __builtin_popcount(hits_imask & ~(0xffffffffu << slot));
            if (STATS) { ++n_nodes; count_wave_step(w_nodes); hist_node_step(ANY, base + rel, oct4); }
            uint32_t key0 = 0u;
            if (UNI && node_step_is_uniform(base + rel, oct4, key0)) {
                CRT_MARK("uninode_begin");
                if (STATS && n_uni) ++*n_uni;
                uint4 n0, n1, n2, n3, n4;
                load_node_scalar(nodes, key0 >> 3, n0, n1, n2, n3, n4);
                const uint32_t oct0 = key0 & 7u;
                const uint32_t hitmask = node8_intersect(n0, n1, n2, n3, n4, o, inv, (oct0 & 4u) == 0u, (oct0 & 2u) == 0u, (oct0 & 1u) == 0u, oct0 * 0x01010101u, max_t);
                cur.x = n1.x;
                tg.x = n1.y;
                cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
                tg.y = hitmask & 0x00ffffffu;
                CRT_MARK("uninode_end");
            } else {
            const uint4* np = node_rows(nodes, base + rel);
            const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
            const uint32_t hitmask = node8_intersect(n0, n1, n2, n3, n4, o, inv, negx, negy, negz, oct4, max_t);
            cur.x = n1.x;
            tg.x = n1.y;
            cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
            tg.y = hitmask & 0x00ffffffu;
            }
            CRT_MARK("node_end");
        } else {
            tg = cur;
            cur = make_uint2(0u, 0u);
        }
        while (tg.y) {
            CRT_MARK("tri_begin");
            const int b = 31 - __builtin_clz(tg.y);
            tg.y &= ~(1u << b);
            const uint32_t ti = tg.x + (uint32_t)b;
            const float4* tp = tri_rows(tris, ti);
            const float4 ta = tp[0], tb = tp[1], tc = tp[2];
            if (STATS) { ++n_tris; count_wave_step(w_tris); }
            float u, v, t;
            if (mt_test(ta, tb, tc, o, d, u, v, t)) {
                if (ANY) {
                    if (t < max_t) { best.tri = (int)ti; CRT_MARK("tri_end"); CRT_MARK("loop_end"); return true; }
                } else {
                    const int id = __float_as_int(ta.w);
                    // SURVEY appendix C tie rule: nearer wins, equal t -> lower original id
                    if (t < best.t || (t == best.t && best.tri >= 0 && id < best.id)) {
                        best.t = t; best.u = u; best.v = v; best.tri = (int)ti; best.id = id;
                        max_t = t;
                    }
                }
            }
            CRT_MARK("tri_end");
        }
        if (!(cur.y & 0xff000000u)) {
            if (sp == 0) break;
            --sp;
            cur = stk[sp * 64];
        }
    }
    CRT_MARK("loop_end");
    return best.tri >= 0;
}

// ---- SEVERAL LANES PER RAY, as a wave drains (round 4) ----
// A lock-step batch starts with one ray per lane and ends on its longest rays: on the 1 M-triangle scene 56 % of the closest-hit node
// steps of the bounce segments run with at most 32 of the 64 lanes enabled, 37 % with at most 8 (any-hit: 68 % / 41 %;
// profiles/r04_lane_hist.txt).  The 8 child tests of a node are independent (cwbvh.fs:376-446) and so are the triangle tests of a leaf,
// so walk_batch gives the rays that are still alive MORE LANES: whenever at most half of the lanes are busy the rays are regrouped into
// groups of K = 2, 4, then 8 adjacent lanes.  A regroup copies the ray's state from its old leader lane into the new group's lanes
// (ds_bpermute: origin, direction, best hit, the two pending masks, stack depth, stack column); after that every lane of a group runs the
// same control flow on the same data — lock-step by construction — except that inside a node step lane `sub` tests children
// [sub * 8 / K, (sub + 1) * 8 / K) only (the group's masks are OR-ed with DPP moves: ~116 / 89 / 65 vector instructions per node step for
// K = 2 / 4 / 8 instead of 230), and a leaf's pending triangles are tested side by side.  The traversal stack is LDS anyway: a group keeps
// using its ray's ORIGINAL column (`col`), written by the group's first lane and read by all; the best hit's (u, v, id) live there too,
// and a finished ray's (t, triangle) are fetched from its group's registers by its original lane after the loop.
// Same nodes in the same order, same tests on the same operands; closest hits fold a leaf's candidates by the rule's own total order
// (nearer t, then lower id — the sequential rule's result for any order of arrival), any-hit walks stop at a leaf's first hit in the
// original order: hits, occlusion and per-ray counters keep the oracle's values.
template <int KL> __device__ __forceinline__ uint32_t dpp_xor(uint32_t x, int step) {       // value of the lane `step` away inside the group (step = 1, 2, 4)
    if (step == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);     // quad_perm [1, 0, 3, 2]
    if (step == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);     // quad_perm [2, 3, 0, 1]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);                   // row_half_mirror: lane i <-> 7 - i of every 8
}
template <int KL> __device__ __forceinline__ uint32_t group_or(uint32_t x) {
    if (KL >= 1) x |= dpp_xor<KL>(x, 1);
    if (KL >= 2) x |= dpp_xor<KL>(x, 2);
    if (KL >= 3) x |= dpp_xor<KL>(x, 4);     // the quads are uniform by now: mirroring the 8 lanes pairs the two quads
    return x;
}
// children [sub * PER, (sub + 1) * PER) of the node, PER = 8 >> KL: the arithmetic of node8_intersect on a part of the slots
template <int KL>
__device__ __forceinline__ uint32_t node8_intersect_part(const uint4 n0, const uint4 n1, const uint4 n2, const uint4 n3, const uint4 n4, vec3 o, vec3 inv,
                                                         bool negx, bool negy, bool negz, uint32_t oct4, float max_t, uint32_t sub) {
    constexpr int PER = 8 >> KL;
    const vec3 p = V3(__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z));
    const uint32_t e_imask = n0.w;
    const vec3 adj_inv = V3(__uint_as_float((e_imask & 0xffu) << 23) * inv.x, __uint_as_float(((e_imask >> 8) & 0xffu) << 23) * inv.y,
                            __uint_as_float(((e_imask >> 16) & 0xffu) << 23) * inv.z);
    const vec3 adj_o = (p - o) * inv;
    const bool hi = (sub * (uint32_t)PER) >= 4u;                       // slots 4..7: the second word of every row pair
    const uint32_t sh = ((sub * (uint32_t)PER) & 3u) * 8u;             // first byte of this lane's slots inside that word
    const uint32_t meta4 = (hi ? n1.w : n1.z) >> sh;
    const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
    const uint32_t inner_mask4 = sign_extend_s8x4(is_inner4 << 3);
    const uint32_t bit_index4 = (meta4 ^ (oct4 & inner_mask4)) & 0x1F1F1F1Fu;
    const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
    const uint32_t qlox = (hi ? n2.y : n2.x) >> sh, qhix = (hi ? n2.w : n2.z) >> sh;
    const uint32_t qloy = (hi ? n3.y : n3.x) >> sh, qhiy = (hi ? n3.w : n3.z) >> sh;
    const uint32_t qloz = (hi ? n4.y : n4.x) >> sh, qhiz = (hi ? n4.w : n4.z) >> sh;
    const uint32_t xmin = negx ? qhix : qlox, xmax = negx ? qlox : qhix;
    const uint32_t ymin = negy ? qhiy : qloy, ymax = negy ? qloy : qhiy;
    const uint32_t zmin = negz ? qhiz : qloz, zmax = negz ? qloz : qhiz;
    uint32_t hit_mask = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const float tminx = __builtin_fmaf(ubyte_f(xmin, j), adj_inv.x, adj_o.x);
        const float tminy = __builtin_fmaf(ubyte_f(ymin, j), adj_inv.y, adj_o.y);
        const float tminz = __builtin_fmaf(ubyte_f(zmin, j), adj_inv.z, adj_o.z);
        const float tmaxx = __builtin_fmaf(ubyte_f(xmax, j), adj_inv.x, adj_o.x);
        const float tmaxy = __builtin_fmaf(ubyte_f(ymax, j), adj_inv.y, adj_o.y);
        const float tmaxz = __builtin_fmaf(ubyte_f(zmax, j), adj_inv.z, adj_o.z);
        const float tmin = __builtin_fmaxf(__builtin_fmaxf(tminx, tminy), __builtin_fmaxf(tminz, 0.0f));
        const float tmax = __builtin_fminf(__builtin_fminf(tmaxx, tmaxy), __builtin_fminf(tmaxz, max_t));
        if (tmin <= tmax) {
            const uint32_t child_bits = (child_bits4 >> (8 * j)) & 0xffu;
            const uint32_t bit_index = (bit_index4 >> (8 * j)) & 0xffu;
            hit_mask |= child_bits << bit_index;
        }
    }
    return hit_mask;
}

// One triangle step of a group of K = 1 << KL lanes (K >= 2): lane `sub` tests the sub-th pending triangle from the top of tg.y.
// Returns true when the ray is done (any-hit: occluded).  Closest hits: the group's candidates are reduced to the one the sequential
// rule would end with — the lexicographic minimum of (t, id) — and accepted against the best so far; its lane writes the hit record.
template <int KL, bool ANY, bool STATS>
__device__ __forceinline__ bool group_tri_step(const float4* __restrict__ tris, vec3 o, vec3 d, uint2& tg, uint32_t sub, float& best_t, int& best_tri,
                                               uint2* hit_uv, uint2* hit_it, uint32_t& n_tris) {
    constexpr uint32_t K = 1u << KL;
    const uint32_t lane = threadIdx.x & 63u;
    // The group takes the K bit POSITIONS from the highest pending triangle down — lane `sub` the position top - sub, if a triangle is pending
    // there — instead of the K highest pending triangles: finding "the sub-th set bit" cost ~60 instructions per step (and seven lane masks
    // the compiler kept in spilled SGPRs), the window ten.  The pending bits of a node are runs of 1 - 3 (a leaf's triangles) in the order of
    // its leaves, so a window of eight positions usually holds them all; the tests, their operands and their order (descending position) stay.
    const uint32_t pending = tg.y;                                            // != 0: the caller's has_tri
    const uint32_t top = 31u - (uint32_t)__builtin_clz(pending);
    const uint32_t lo = top >= K - 1u ? top - (K - 1u) : 0u;                  // the window is [lo, top]
    const uint32_t rest = pending & ~(0xffffffffu << lo);
    const uint32_t tested = (uint32_t)__builtin_popcount(pending >> lo);
    const bool mine_set = sub <= top && ((pending >> (top - sub)) & 1u) != 0u;
    bool hit = false;
    float u = 0.f, v = 0.f, t = 0.f;
    int id = 0x7fffffff;
    uint32_t ti = 0;
    if (mine_set) {
        ti = tg.x + (top - sub);
        const float4* tp = tri_rows(tris, ti);
        const float4 ta = tp[0], tb = tp[1], tc = tp[2];
        hit = mt_test(ta, tb, tc, o, d, u, v, t);
        id = __float_as_int(ta.w);
    }
    tg.y = rest;
    if (ANY) {
        const uint32_t mine = (uint32_t)(__ballot(hit && t < best_t) >> (lane & ~(K - 1u))) & ((1u << K) - 1u);      // my group's hits, bit = sub
        // tests up to the first hit in the original order: the pending triangles at and above its position
        if (STATS && sub == 0u) n_tris += mine ? (uint32_t)__builtin_popcount(pending >> (top - (uint32_t)__builtin_ctz(mine))) : tested;
        if (mine) { best_tri = (int)tg.x; tg.y = 0u; return true; }
        return false;
    }
    if (STATS && sub == 0u) n_tris += tested;
    // candidate of this lane: (t, id, triangle), or (+inf, max) when it has none
    uint32_t ct = hit ? __float_as_uint(t) : 0x7f800000u, ci = hit ? (uint32_t)id : 0x7fffffffu, cx = ti;
#pragma unroll
    for (int step = 1; step < (int)K; step <<= 1) {
        const uint32_t pt = dpp_xor<KL>(ct, step), pi = dpp_xor<KL>(ci, step), px = dpp_xor<KL>(cx, step);
        const float a = __uint_as_float(ct), b = __uint_as_float(pt);
        const bool other = b < a || (b == a && (int)pi < (int)ci);      // t >= 0 for every hit: plain float order; equal t -> lower id
        ct = other ? pt : ct; ci = other ? pi : ci; cx = other ? px : cx;
    }
    const float wt = __uint_as_float(ct);
    if (ct != 0x7f800000u) {
        bool take = wt < best_t;
        if (wt == best_t && best_tri >= 0) take = (int)ci < (int)hit_it->x;
        if (take) {
            best_t = wt; best_tri = (int)cx;
            if (hit && ti == cx) { *hit_uv = make_uint2(__float_as_uint(u), __float_as_uint(v)); hit_it->x = ci; }      // the winner's own lane has (u, v)
        }
    }
    return false;
}

// The group phase of walk_batch (and of the first segment's shadow walk): `busy` lanes hand their rays — at most 64 >> GROUP_KL of them —
// to groups of K adjacent lanes, which finish them; on return out.t / out.tri of a lane whose ray moved hold its result (closest hits: the
// (u, v, id) record is in the lane's own column slots as always).  Every lane of the wave calls this together.
template <bool ANY, bool STATS, bool UNIFORM_O>
__device__ __forceinline__ void group_phase(const uint4* __restrict__ nodes, const float4* __restrict__ tris, uint2* base, int stack_entries, uint32_t* overflow,
                                            bool busy, vec3 o_lane, vec3 d, float best_t, int best_tri, uint2 cur, uint2 tg, int sp, uint32_t tri_min,
                                            HitState& out, uint32_t& n_nodes, uint32_t& n_tris, uint32_t& w_nodes, uint32_t& w_tris, vec3 o_uniform) {
    constexpr uint32_t KL = GROUP_KL, K = 1u << KL;
    const uint32_t lane = threadIdx.x & 63u;
    uint2* const slot_uv = base + stack_entries * 64;
    uint2* const slot_it = base + (stack_entries + 1) * 64;
    vec3 dc; bool negx, negy, negz; uint32_t oct4; vec3 inv;
    const bool moved = busy;                        // this lane's ray is finished by a group: its result comes back from the group's registers
    {
        // ---------------- regroup: each of the <= 64 / K rays still alive gets K adjacent lanes ----------------
        CRT_MARK("loop_begin regroup");
        const unsigned long long lm = __ballot(busy);
        const uint32_t n_rays = (uint32_t)__builtin_popcountll(lm);
        const uint32_t my_group = (uint32_t)__builtin_popcountll(lm & ((1ull << lane) - 1ull));      // of a lane whose ray moves: the group that takes it
        if (busy) slot_it[my_group].y = lane;
        __builtin_amdgcn_wave_barrier();
        const uint32_t g = lane >> KL, sub = lane & (K - 1u);
        const bool act = g < n_rays;
        const int src = act ? (int)slot_it[g].y : (int)lane;
        if (!UNIFORM_O) o_lane = V3(__shfl(o_lane.x, src), __shfl(o_lane.y, src), __shfl(o_lane.z, src));
        d = V3(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
        best_t = __shfl(best_t, src); best_tri = __shfl(best_tri, src);
        cur.x = (uint32_t)__shfl((int)cur.x, src); cur.y = (uint32_t)__shfl((int)cur.y, src);
        tg.x = (uint32_t)__shfl((int)tg.x, src); tg.y = (uint32_t)__shfl((int)tg.y, src);
        sp = __shfl(sp, src);
        const uint32_t col = (uint32_t)src;         // the ray's original lane: its stack column and hit-record slots
        __builtin_amdgcn_wave_barrier();
        busy = act;
        dc = V3(clamp_dir(d.x), clamp_dir(d.y), clamp_dir(d.z));
        negx = dc.x < 0.0f; negy = dc.y < 0.0f; negz = dc.z < 0.0f;
        oct4 = (negx ? 0u : 0x04040404u) | (negy ? 0u : 0x02020202u) | (negz ? 0u : 0x01010101u);
        inv = V3(rcp_ieee(dc.x), rcp_ieee(dc.y), rcp_ieee(dc.z));
        CRT_MARK("loop_end");
        // ---------------- phase 2: K lanes per ray ----------------
        uint2* const stk = base + col;
        const vec3 o = UNIFORM_O ? o_uniform : o_lane;
        CRT_MARK("loop_begin lanes2");
        if (!busy) { cur.y = 0u; tg.y = 0u; sp = 0; }       // lanes outside the groups: nothing pending (see walk_batch: no flag is carried through the loop)
        for (;;) {
            const bool has_tri = tg.y != 0u;
            const bool can_node = !has_tri && (cur.y & 0xff000000u);
            const unsigned long long m_tri = __ballot(has_tri), m_node = __ballot((cur.y & 0xff000000u) != 0u) & ~m_tri;
            if ((m_tri | m_node) == 0ull) break;
            const uint32_t n_tri = (uint32_t)__builtin_popcountll(m_tri), n_node = (uint32_t)__builtin_popcountll(m_node);
            const bool node_phase = n_node != 0u && n_node >= tri_min * n_tri;      // both sides count lanes, i.e. rays x K
            bool finished = false;
            if (node_phase) {
                if (can_node) {
                    CRT_MARK("node_begin");
                    const uint32_t hits_imask = cur.y;
                    const int off = 31 - __builtin_clz(hits_imask);
                    const uint32_t nbase = cur.x;
                    cur.y &= ~(1u << off);
                    if (cur.y & 0xff000000u) {
                        if (sp < stack_entries) { if (sub == 0u) stk[sp * 64] = cur; ++sp; } else if (sub == 0u) atomicAdd(overflow, 1u);
                    }
                    const uint32_t slot = (uint32_t)(off - 24) ^ (oct4 & 0xffu);
                    const uint32_t nidx = nbase + (uint32_t)__builtin_popcount(hits_imask & ~(0xffffffffu << slot));
                    const uint4* np = node_rows(nodes, nidx);
                    // a lane of the group fetches only the words its child's bytes sit in: 52 instead of 80 bytes per lane through the
                    // texture-address path (+0.7 .. 1.1 % on the multi-segment frames; the eight lanes of a group ask for the same node)
                    const uint32_t* nw = reinterpret_cast<const uint32_t*>(np) + (((sub * (uint32_t)(8 >> KL)) >= 4u) ? 1u : 0u);
                    const uint4 n0 = np[0];
                    const uint2 n1xy = *reinterpret_cast<const uint2*>(np + 1);
                    const uint32_t mw = nw[6], w2l = nw[8], w2h = nw[10], w3l = nw[12], w3h = nw[14], w4l = nw[16], w4h = nw[18];
                    const uint4 n1 = make_uint4(n1xy.x, n1xy.y, mw, mw), n2 = make_uint4(w2l, w2l, w2h, w2h), n3 = make_uint4(w3l, w3l, w3h, w3h), n4 = make_uint4(w4l, w4l, w4h, w4h);
                    if (STATS) { if (sub == 0u) ++n_nodes; count_wave_step(w_nodes); }
                    const uint32_t hitmask = group_or<KL>(node8_intersect_part<KL>(n0, n1, n2, n3, n4, o, inv, negx, negy, negz, oct4, best_t, sub));
                    cur.x = n1.x;
                    tg.x = n1.y;
                    cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
                    tg.y = hitmask & 0x00ffffffu;
                    CRT_MARK("node_end");
                }
            } else if (has_tri) {
                CRT_MARK("tri_begin");
                if (STATS) count_wave_step(w_tris);
                finished = group_tri_step<KL, ANY, STATS>(tris, o, d, tg, sub, best_t, best_tri, slot_uv + col, slot_it + col, n_tris);
                CRT_MARK("tri_end");
            }
            if (!finished && tg.y == 0u && !(cur.y & 0xff000000u) && sp != 0) { --sp; cur = stk[sp * 64]; }
            if (ANY && finished) { cur.y = 0u; sp = 0; }    // the group's lanes keep the ray's result in their registers
        }
        CRT_MARK("loop_end");
        // a ray that moved fetches its result from the first lane of its group
        const float t_back = __shfl(best_t, (int)(my_group << KL));
        const int tri_back = __shfl(best_tri, (int)(my_group << KL));
        if (moved) { out.t = t_back; out.tri = tri_back; }
        __builtin_amdgcn_wave_barrier();            // the hit records were written by other lanes
    }
}

// `base` = the wave's LDS region (lane 0's stack column); every lane of the wave calls this together.  On return `out` holds, in every
// lane that had a ray, its closest hit (ANY: out.tri >= 0 means occluded).  max_kl = 0: one lane per ray throughout.
// Two loops: phase 1 is the lock-step voting loop (walk_pool's, without the refill), one ray per lane, and ends when at most 64 >> GROUP_KL rays are
// left; those are regrouped and finished by phase 2, which knows nothing but groups.  (One loop that carried the group size as a variable
// cost the one-lane phase 5 %: a guard on every stack write, a switch in every step.)
template <bool ANY, bool STATS, bool UNIFORM_O, bool UNI = false>
__device__ __forceinline__ void walk_batch(const uint4* __restrict__ nodes, const float4* __restrict__ tris, uint2* base, int stack_entries, uint32_t* overflow,
                                           bool has_ray, vec3 o_in, vec3 d, float tmax_in, uint32_t tri_min, uint32_t max_kl, HitState& out,
                                           uint32_t& n_nodes, uint32_t& n_tris, uint32_t& w_nodes, uint32_t& w_tris, vec3 o_uniform = V3(0.f, 0.f, 0.f),
                                           uint32_t* n_uni = nullptr, const float4* planes = nullptr) {
    constexpr uint32_t KL = GROUP_KL;
    const uint32_t lane = threadIdx.x & 63u;
    uint2* const slot_uv = base + stack_entries * 64;              // [col] (u, v) of the best hit
    uint2* const slot_it = base + (stack_entries + 1) * 64;        // [col] (original id of the best hit, regroup scratch)
    vec3 o_lane = o_in;
    float best_t = tmax_in;
    int best_tri = -1;
    vec3 oo = UNIFORM_O ? o_uniform : o_lane;
    bool busy = has_ray && __builtin_isfinite(oo.x) && __builtin_isfinite(oo.y) && __builtin_isfinite(oo.z);
    vec3 dc = V3(clamp_dir(d.x), clamp_dir(d.y), clamp_dir(d.z));
    bool negx = dc.x < 0.0f, negy = dc.y < 0.0f, negz = dc.z < 0.0f;
    uint32_t oct4 = (negx ? 0u : 0x04040404u) | (negy ? 0u : 0x02020202u) | (negz ? 0u : 0x01010101u);
    vec3 inv = V3(rcp_ieee(dc.x), rcp_ieee(dc.y), rcp_ieee(dc.z));
    uint2 cur = busy ? make_uint2(0u, 0x80000000u) : make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    int sp = 0;
    bool regroup = false;
    {
        // ---------------- phase 1: one ray per lane ----------------
        uint2* const stk = base + lane;
        const vec3 o = UNIFORM_O ? o_uniform : o_lane;
        CRT_MARK("loop_begin lanes1");
        for (;;) {
            // No flag is carried from one iteration to the next: a lane has a ray exactly while it has a triangle group or inner hits
            // pending (what is left of a finished ray is cleared below), so the three counts come from the two registers the step tests
            // anyway.  (A loop-carried bool costs a v_cndmask + v_cmp pair at every ballot: the compiler materialises the lane mask.)
            const bool has_tri = tg.y != 0u;
            const bool can_node = !has_tri && (cur.y & 0xff000000u);
            const unsigned long long m_tri = __ballot(has_tri), m_node = __ballot((cur.y & 0xff000000u) != 0u) & ~m_tri;
            const uint32_t n_tri = (uint32_t)__builtin_popcountll(m_tri), n_node = (uint32_t)__builtin_popcountll(m_node);
            const uint32_t n_busy = n_tri + n_node;
            if (n_busy == 0u) break;
            if (max_kl != 0u && n_busy <= (64u >> KL)) { regroup = true; break; }
            const bool node_phase = n_node != 0u && n_node >= tri_min * n_tri;      // the vote of walk_pool
            bool finished = false;
            if (node_phase) {
                if (can_node) {
                    CRT_MARK("node_begin");
                    const uint32_t hits_imask = cur.y;
                    const int off = 31 - __builtin_clz(hits_imask);
                    const uint32_t nbase = cur.x;
                    cur.y &= ~(1u << off);
                    if (cur.y & 0xff000000u) { if (sp < stack_entries) { stk[sp * 64] = cur; ++sp; } else atomicAdd(overflow, 1u); }
                    const uint32_t slot = (uint32_t)(off - 24) ^ (oct4 & 0xffu);
                    const uint32_t nidx = nbase + (uint32_t)__builtin_popcount(hits_imask & ~(0xffffffffu << slot));
                    if (STATS) { ++n_nodes; count_wave_step(w_nodes); hist_node_step(ANY, nidx, oct4); }
                    uint32_t key0 = 0u;
                    if (UNI && node_step_is_uniform(nidx, oct4, key0)) {
                        // every enabled lane asks for this node and shares the octant: the node comes through the scalar cache
                        CRT_MARK("uninode_begin");
                if (STATS && n_uni) ++*n_uni;
                        uint4 n0, n1;
                        const uint32_t oct0 = key0 & 7u;
                        // (the host builds the float planes for every scene: finish_scene_setup)
                        const uint32_t hitmask = node8_intersect_planes(n0, n1, nodes, planes, key0 >> 3, o, inv, oct0, best_t);
                        cur.x = n1.x;
                        tg.x = n1.y;
                        cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
                        tg.y = hitmask & 0x00ffffffu;
                        CRT_MARK("uninode_end");
                    } else {
                    const uint4* np = node_rows(nodes, nidx);
                    const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
                    const uint32_t hitmask = node8_intersect(n0, n1, n2, n3, n4, o, inv, negx, negy, negz, oct4, best_t);
                    cur.x = n1.x;
                    tg.x = n1.y;
                    cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
                    tg.y = hitmask & 0x00ffffffu;
                    }
                    CRT_MARK("node_end");
                }
            } else if (has_tri) {
                CRT_MARK("tri_begin");
                const int b = 31 - __builtin_clz(tg.y);
                tg.y &= ~(1u << b);
                const uint32_t ti = tg.x + (uint32_t)b;
                const float4* tp = tri_rows(tris, ti);
                const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                if (STATS) { ++n_tris; count_wave_step(w_tris); }
                float u, v, t;
                if (mt_test(ta, tb, tc, o, d, u, v, t)) {
                    if (ANY) {
                        if (t < best_t) { best_tri = (int)ti; finished = true; tg.y = 0u; }
                    } else {
                        const int id = __float_as_int(ta.w);
                        bool take = t < best_t;
                        if (t == best_t && best_tri >= 0) take = id < (int)slot_it[lane].x;
                        if (take) { best_t = t; best_tri = (int)ti; slot_uv[lane] = make_uint2(__float_as_uint(u), __float_as_uint(v)); slot_it[lane].x = (uint32_t)id; }
                    }
                }
                CRT_MARK("tri_end");
            }
            // a lane with nothing pending pops its stack; one without a ray has an empty stack and nothing happens to it
            if (!finished && tg.y == 0u && !(cur.y & 0xff000000u) && sp != 0) { --sp; cur = stk[sp * 64]; }
            if (ANY && finished) { cur.y = 0u; sp = 0; }      // an occluded ray leaves inner hits and stack entries behind (its result stays in this lane's registers)
        }
        busy = tg.y != 0u || (cur.y & 0xff000000u);
        CRT_MARK("loop_end");
    }
    out.t = best_t; out.tri = best_tri;
    if (regroup)
        group_phase<ANY, STATS, UNIFORM_O>(nodes, tris, base, stack_entries, overflow, busy, o_lane, d, best_t, best_tri, cur, tg, sp, tri_min, out,
                                           n_nodes, n_tris, w_nodes, w_tris, o_uniform);
    out.u = 0.f; out.v = 0.f; out.id = -1;
    if (!ANY && out.tri >= 0) {
        const uint2 uv = slot_uv[lane];
        out.u = __uint_as_float(uv.x); out.v = __uint_as_float(uv.y); out.id = (int)slot_it[lane].x;
    }
    __builtin_amdgcn_wave_barrier();              // the slots are free again
}

// Wave-level traversal of a POOL of rays with lane refill (persistent threads in the sense of Aila & Laine): a lane whose ray has finished
// takes the next ray of the wave's pool instead of idling until the slowest ray of its 64-ray batch is done, and when the pool has run dry
// the wave's last eight rays get eight lanes each (group_phase, max_kl != 0).  Per-ray work and its order are unchanged, so hits and visit
// counters stay bit-identical to the oracle.  The pool is [pool_begin, pool_end); refill happens when at least `refill_min` lanes are idle
// (or none is busy); refill_min = 65 makes it one lock-step batch per 64 rays of the pool.
//   bool load(idx, o, d, tmax)  fetches ray idx of the pool (false: no ray in this slot);  done(idx, best, hit)  consumes its result.
// The best hit's (u, v) and original id live in the lane's hit slots (rt_kernels.hpp CRT_HIT_SLOTS), as in walk_batch.  `base` = the wave's
// LDS region (lane 0's stack column); every lane of the wave calls this together.  Per-RAY counters (k_trace's stats) need max_kl = 0: a
// regrouped ray's visits are counted by its group's first lane.
#define CRT_COUNTER_STRIDE 32u   // uint32 slots between two per-group counters (128 B)
// Where the rays of a walk_pool come from (all members wave-uniform).
//   PoolStatic: the wave's own slice [next, end) of an index space.
//   PoolStream (PERSISTENT THREADS, round 5): the wave draws from queues shared by the whole grid — `n_queues` sub-queues, queue k holding
//   counts[(k >> 3) * count_stride + (k & 7) * CRT_COUNTER_STRIDE] rays at flat indices k * sub_capacity + e — through one cursor per queue
//   (cursors[k * CRT_COUNTER_STRIDE], zero at launch; one returning atomic per refill, by one lane).  A wave starts on the queue of its own XCD
//   group and moves on to the next queue when one runs dry, so no wave idles while any queue holds rays and the grid ends on ONE drain phase
//   instead of one per pool: the launch is as many workgroups as the chip holds waves, each alive until every queue is empty.
struct PoolStatic {
    uint32_t next, end;
    __device__ __forceinline__ bool more() const { return next < end; }
    __device__ __forceinline__ uint32_t grab(uint32_t want, uint32_t& got) {
        const uint32_t b = next;
        got = end - next < want ? end - next : want;
        next += got;
        return b;
    }
};
struct PoolStream {
    const uint32_t* counts; uint32_t* cursors;
    uint32_t count_stride, sub_capacity, n_queues, q, tried, chunk;      // sub_capacity: flat-index distance between two queues
    uint32_t next, end;          // what is left of the chunk this wave has reserved (flat indices)
    __device__ __forceinline__ bool more() const { return next < end || tried < n_queues; }
    __device__ __forceinline__ uint32_t grab(uint32_t want, uint32_t& got) {
        // a refill hands out rays of the wave's reserved chunk; a new chunk costs one returning atomic (its round trip is on the wave's
        // critical path: one per `chunk` rays, not one per refill — per-refill atomics lost 8 - 25 % on four segments of the 1 M-triangle scene)
        while (next == end && tried < n_queues) {
            const uint32_t n = counts[(size_t)(q >> 3) * count_stride + (q & 7u) * CRT_COUNTER_STRIDE];
            const int leader = __builtin_ctzll(__ballot(true));
            uint32_t b = 0;
            if ((int)(threadIdx.x & 63u) == leader) b = atomicAdd(cursors + q * CRT_COUNTER_STRIDE, chunk);
            b = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)b, leader));
            if (b < n) { next = q * sub_capacity + b; end = next + (n - b < chunk ? n - b : chunk); break; }
            q = q + 1u == n_queues ? 0u : q + 1u;        // this queue is dry: on to the next
            ++tried;
        }
        const uint32_t first = next;
        got = end - next < want ? end - next : want;
        next += got;
        return first;
    }
};

template <bool ANY, bool STATS, typename Source, typename Load, typename Done>
__device__ __forceinline__ void walk_pool(const uint4* __restrict__ nodes, const float4* __restrict__ tris, uint2* base, int stack_entries, uint32_t* overflow,
                                          Source src, uint32_t refill_min, uint32_t tri_min, uint32_t max_kl, Load load, Done done,
                                          uint32_t& n_nodes, uint32_t& n_tris, uint32_t& w_nodes, uint32_t& w_tris) {
    const uint32_t lane = threadIdx.x & 63u;
    uint2* const stk = base + lane;
    uint32_t idx = 0;
    vec3 o = V3(0.f, 0.f, 0.f), d = V3(0.f, 0.f, 1.f), inv = V3(0.f, 0.f, 0.f);
    bool negx = false, negy = false, negz = false;
    uint32_t oct4 = 0;
    float best_t = 0.f;                             // closest hit so far (closest-hit walks: also the far clip of every box and triangle test)
    int best_tri = -1;
    uint2* const hit_uv = stk + stack_entries * 64;             // CRT_HIT_SLOTS
    uint2* const hit_id = stk + (stack_entries + 1) * 64;
    int sp = 0;
    uint2 cur = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    bool regroup = false;
    auto finish = [&](float t, int tri) {
        HitState best;
        best.t = t; best.tri = tri; best.u = 0.f; best.v = 0.f; best.id = -1;
        if (!ANY && tri >= 0) {
            const uint2 uv = *hit_uv;
            best.u = __uint_as_float(uv.x); best.v = __uint_as_float(uv.y); best.id = (int)hit_id->x;
        }
        done(idx, best, tri >= 0);
    };
    CRT_MARK("loop_begin pool");
    for (;;) {
        // no flag carried through the loop (see walk_batch): a lane has a ray exactly while it has a triangle group or inner hits pending; a
        // ray that is loaded without either (non-finite origin, empty slot) is finished by the end of the same iteration
        bool busy = tg.y != 0u || (cur.y & 0xff000000u) != 0u;
        if (src.more()) {
            const unsigned long long idle = __ballot(!busy);
            const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
            if (n_idle >= refill_min || n_idle == 64u) {
                uint32_t got;
                const uint32_t first = src.grab(n_idle, got);
                const uint32_t rank = (uint32_t)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
                if (!busy && rank < got) {
                    idx = first + rank;
                    float tmax_in;
                    const bool has_ray = load(idx, o, d, tmax_in);
                    best_t = tmax_in; best_tri = -1;
                    sp = 0;
                    busy = true;
                    // same prologue as traverse(): non-finite origin -> immediate miss; clamp zero direction components
                    const bool finite = __builtin_isfinite(o.x) && __builtin_isfinite(o.y) && __builtin_isfinite(o.z);
                    const vec3 dc = V3(clamp_dir(d.x), clamp_dir(d.y), clamp_dir(d.z));
                    negx = dc.x < 0.0f; negy = dc.y < 0.0f; negz = dc.z < 0.0f;
                    oct4 = (negx ? 0u : 0x04040404u) | (negy ? 0u : 0x02020202u) | (negz ? 0u : 0x01010101u);
                    inv = V3(rcp_ieee(dc.x), rcp_ieee(dc.y), rcp_ieee(dc.z));
                    cur = (finite && has_ray) ? make_uint2(0u, 0x80000000u) : make_uint2(0u, 0u);
                    tg = make_uint2(0u, 0u);
                }
            }
        }
        const unsigned long long m_busy = __ballot(busy);
        if (m_busy == 0ull) break;                  // pool drained and every lane finished
        // the pool has run dry and at most eight rays are left: they get eight lanes each (a just-loaded ray with nothing pending is
        // finished by this iteration first: the group phase wants rays that have something pending)
        if (max_kl != 0u && !src.more() && (uint32_t)__builtin_popcountll(m_busy) <= (64u >> GROUP_KL) &&
            __ballot(busy && tg.y == 0u && !(cur.y & 0xff000000u)) == 0ull) { regroup = true; break; }

        // ---- one step per iteration: either a node step (lanes with inner hits pending and no triangle group pending) or a triangle step
        // (lanes with a triangle group pending).  The wave votes: triangle tests are postponed while node-ready lanes outnumber waiting
        // lanes tri_min : 1, so that the Moller-Trumbore block and the node block each run with more lanes enabled.  A lane's own sequence
        // of node fetches and triangle tests is unchanged (it cannot fetch a node while its triangle group is pending).
        // (A wave-level dedup of the node loads — one leader lane per distinct node, followers fed by ds_bpermute — was measured and
        // dropped: 0.469 vs 0.394 ms on coherent primary rays, 0.509 vs 0.434 ms on bounce rays: the bound is VALU issue.)
        const bool has_tri = busy && tg.y != 0u;
        const bool can_node = busy && !has_tri && (cur.y & 0xff000000u);
        const uint32_t n_tri = (uint32_t)__builtin_popcountll(__ballot(has_tri));
        const uint32_t n_node = (uint32_t)__builtin_popcountll(__ballot(can_node));
        const bool node_phase = n_node != 0u && n_node >= tri_min * n_tri;
        bool finished = false;
        if (node_phase) {
            if (can_node) {
                CRT_MARK("node_begin");
                const uint32_t hits_imask = cur.y;
                const int off = 31 - __builtin_clz(hits_imask);
                const uint32_t nbase = cur.x;
                cur.y &= ~(1u << off);
                // crt_scene_create sizes the stack from the validated depth of the tree, so a push always fits; if a caller-supplied tree
                // ever got past the validator the dropped push is counted, not silent
                if (cur.y & 0xff000000u) { if (sp < stack_entries) { stk[sp * 64] = cur; ++sp; } else atomicAdd(overflow, 1u); }
                const uint32_t slot = (uint32_t)(off - 24) ^ (oct4 & 0xffu);
                const uint32_t nidx = nbase + (uint32_t)__builtin_popcount(hits_imask & ~(0xffffffffu << slot));
                const uint4* np = node_rows(nodes, nidx);
                const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
                if (STATS) { ++n_nodes; count_wave_step(w_nodes); hist_node_step(ANY, nidx, oct4); }
                const uint32_t hitmask = node8_intersect(n0, n1, n2, n3, n4, o, inv, negx, negy, negz, oct4, best_t);
                cur.x = n1.x;
                tg.x = n1.y;
                cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
                tg.y = hitmask & 0x00ffffffu;
                CRT_MARK("node_end");
            }
        } else if (has_tri) {
            CRT_MARK("tri_begin");
            const int b = 31 - __builtin_clz(tg.y);
            tg.y &= ~(1u << b);
            const uint32_t ti = tg.x + (uint32_t)b;
            const float4* tp = tri_rows(tris, ti);
            const float4 ta = tp[0], tb = tp[1], tc = tp[2];
            if (STATS) { ++n_tris; count_wave_step(w_tris); }
            float u, v, t;
            if (mt_test(ta, tb, tc, o, d, u, v, t)) {
                if (ANY) {
                    if (t < best_t) { best_tri = (int)ti; finished = true; tg.y = 0u; }
                } else {
                    // nearer wins; equal t -> lower original id; the stored id is fetched only when t ties
                    const int id = __float_as_int(ta.w);
                    bool take = t < best_t;
                    if (t == best_t && best_tri >= 0) take = id < (int)hit_id->x;
                    if (take) { best_t = t; best_tri = (int)ti; *hit_uv = make_uint2(__float_as_uint(u), __float_as_uint(v)); hit_id->x = (uint32_t)id; }
                }
            }
            CRT_MARK("tri_end");
        }
        // a lane with neither a triangle group nor inner hits left pops its stack, or is done
        if (busy && !finished && tg.y == 0u && !(cur.y & 0xff000000u)) {
            if (sp == 0) finished = true;
            else { --sp; cur = stk[sp * 64]; }
        }
        if (finished) {
            finish(best_t, best_tri);
            cur.y = 0u; sp = 0;                     // what an occluded ray leaves behind
        }
    }
    CRT_MARK("loop_end");
    if (regroup) {
        const bool moved = tg.y != 0u || (cur.y & 0xff000000u) != 0u;
        HitState out;
        out.t = best_t; out.tri = best_tri;
        group_phase<ANY, STATS, false>(nodes, tris, base, stack_entries, overflow, moved, o, d, best_t, best_tri, cur, tg, sp, tri_min, out, n_nodes, n_tris, w_nodes, w_tris,
                                       V3(0.f, 0.f, 0.f));
        if (moved) finish(out.t, out.tri);
    }
}

// Any-hit walk of one batch: the plain per-lane loop of traverse<true> (each lane tests its leaf's triangles right after the node that found
// them — the fastest form for the coherent shadow rays of primary hits) until at most 64 >> GROUP_KL rays are left, then the group phase.
// Every lane of the wave calls this together; returns whether this lane's ray is occluded.
template <bool STATS, bool UNI = false>
__device__ __forceinline__ bool traverse_any_then_groups(const uint4* __restrict__ nodes, const float4* __restrict__ tris, uint2* base, int stack_entries,
                                                         uint32_t* overflow, bool has_ray, vec3 o, vec3 d, float tmax, uint32_t tri_min, uint32_t max_kl,
                                                         uint32_t& n_nodes, uint32_t& n_tris, uint32_t& w_nodes, uint32_t& w_tris, uint32_t* n_uni = nullptr,
                                                         const float4* planes = nullptr) {
    constexpr uint32_t KL = GROUP_KL;
    const uint32_t lane = threadIdx.x & 63u;
    uint2* const stk = base + lane;
    bool busy = has_ray && __builtin_isfinite(o.x) && __builtin_isfinite(o.y) && __builtin_isfinite(o.z);
    const vec3 dc = V3(clamp_dir(d.x), clamp_dir(d.y), clamp_dir(d.z));
    const bool negx = dc.x < 0.0f, negy = dc.y < 0.0f, negz = dc.z < 0.0f;
    const uint32_t oct4 = (negx ? 0u : 0x04040404u) | (negy ? 0u : 0x02020202u) | (negz ? 0u : 0x01010101u);
    const vec3 inv = V3(rcp_ieee(dc.x), rcp_ieee(dc.y), rcp_ieee(dc.z));
    uint2 cur = busy ? make_uint2(0u, 0x80000000u) : make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    int sp = 0, hit_tri = -1;
    bool regroup = false;
    CRT_MARK("loop_begin plain");
    for (;;) {
        // no flag carried through the loop (see walk_batch): a lane has a ray exactly while inner hits are pending — a leaf's triangles are
        // tested in the iteration that found them, and what an occluded ray leaves behind is cleared
        const bool busy_now = (cur.y & 0xff000000u) != 0u;
        const uint32_t n_busy = (uint32_t)__builtin_popcountll(__ballot(busy_now));
        if (n_busy == 0u) break;
        if (max_kl != 0u && n_busy <= (64u >> KL)) { regroup = true; break; }
        if (busy_now) {
            {
                CRT_MARK("node_begin");
                const uint32_t hits_imask = cur.y;
                const int off = 31 - __builtin_clz(hits_imask);
                const uint32_t nbase = cur.x;
                cur.y &= ~(1u << off);
                if (cur.y & 0xff000000u) { if (sp < stack_entries) { stk[sp * 64] = cur; ++sp; } else atomicAdd(overflow, 1u); }
                const uint32_t slot = (uint32_t)(off - 24) ^ (oct4 & 0xffu);
                const uint32_t rel = __builtin_popcount(hits_imask & ~(0xffffffffu << slot));
                if (STATS) { ++n_nodes; count_wave_step(w_nodes); hist_node_step(true, nbase + rel, oct4); }
                uint32_t key0 = 0u;
                if (UNI && node_step_is_uniform(nbase + rel, oct4, key0)) {
                    CRT_MARK("uninode_begin");
                if (STATS && n_uni) ++*n_uni;
                    uint4 n0, n1;
                    const uint32_t oct0 = key0 & 7u;
                    const uint32_t hitmask = node8_intersect_planes(n0, n1, nodes, planes, key0 >> 3, o, inv, oct0, tmax);
                    cur.x = n1.x;
                    tg.x = n1.y;
                    cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
                    tg.y = hitmask & 0x00ffffffu;
                    CRT_MARK("uninode_end");
                } else {
                const uint4* np = node_rows(nodes, nbase + rel);
                const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
                const uint32_t hitmask = node8_intersect(n0, n1, n2, n3, n4, o, inv, negx, negy, negz, oct4, tmax);
                cur.x = n1.x;
                tg.x = n1.y;
                cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
                tg.y = hitmask & 0x00ffffffu;
                }
                CRT_MARK("node_end");
            }
            while (tg.y) {
                CRT_MARK("tri_begin");
                const int b = 31 - __builtin_clz(tg.y);
                tg.y &= ~(1u << b);
                const uint32_t ti = tg.x + (uint32_t)b;
                const float4* tp = tri_rows(tris, ti);
                const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                if (STATS) { ++n_tris; count_wave_step(w_tris); }
                float u, v, t;
                if (mt_test(ta, tb, tc, o, d, u, v, t) && t < tmax) { hit_tri = (int)ti; tg.y = 0u; cur.y = 0u; sp = 0; }
                CRT_MARK("tri_end");
            }
            if (!(cur.y & 0xff000000u) && sp != 0) { --sp; cur = stk[sp * 64]; }
        }
    }
    busy = (cur.y & 0xff000000u) != 0u;
    CRT_MARK("loop_end");
    HitState out;
    out.t = tmax; out.tri = hit_tri;
    if (regroup)
        group_phase<true, STATS, false>(nodes, tris, base, stack_entries, overflow, busy, o, d, tmax, -1, cur, tg, sp, tri_min, out, n_nodes, n_tris, w_nodes, w_tris,
                                        V3(0.f, 0.f, 0.f));
    return out.tri >= 0;
}

// ------------------------------------------------------------------ scheduling -------

// XCD-aware work distribution shared by all traversal kernels.  By default the host launches one (single-wave)
// workgroup per batch, so `it` below only ever takes the value 0 and the hardware dispatcher balances the load; with a
// smaller, persistent grid (option "oversubscribe" >= 1) the same mapping is a static round-robin schedule.
//
// Work is cut into 64-ray batches (one wave; one 8x8 pixel block for primary rays) and chunks of 4
// batches (one workgroup pass, a 16x16 pixel patch).  Every chunk belongs to one of 8 *groups*:
//   - dense index spaces (pixels, explicit ray buffers): 64 consecutive batches form a unit (one 64x64
//     tile) and unit u belongs to group u & 7;
//   - device-written queues (path rays, shadow rays) are 8 sub-queues, one per group: a group appends to
//     and later consumes its own sub-queue, so a ray stays with the group (and normally the L2) that
//     already holds its neighbourhood, and the append atomics are spread over 8 counters on separate
//     cache lines (a single counter saturates near 90 returning atomics per microsecond: at one atomic
//     per 64-ray wave that capped a whole 1080p frame at ~0.18 ms; measured 0.183 -> 0.074 ms).
// Workgroups are dealt round-robin to the 8 XCDs by the dispatcher, so blockIdx & 7 labels workgroups
// that share an L2 (a speed heuristic only; correctness never depends on placement).  Workgroup j of a
// group takes that group's chunks j, j + groups_size, ...  A dynamic variant (one returning atomic per
// chunk on a per-group counter, stealing from other groups when drained) was measured and dropped: equal
// on the 1 M-triangle scenes (0.404 vs 0.396 ms) and 1.8x slower on Cornell (0.132 vs 0.074 ms), where the
// per-chunk atomic + two barriers sit on the critical path of very short rays.
// The pass loop of a workgroup over its chunks.  The default build launches one workgroup per chunk (the hardware dispatcher is the
// scheduler), so the loop body runs once — and is written as a loop the compiler can see runs once: around a real loop it hoists
// every wave-uniform value and constant of the body into SGPRs that then live (or spill into VGPR lanes) across the whole kernel.
#ifdef CRT_EXPERIMENTS
#define CRT_CHUNK_LOOP(it) for (uint32_t it = 0;; ++it)          // persistent grids (option "oversubscribe")
#else
#define CRT_CHUNK_LOOP(it) for (uint32_t it = 0; it < 1u; ++it)
#endif
#define CRT_NO_WORK 0xffffffffu

__device__ __forceinline__ uint32_t dense_chunks_of_group(uint32_t n_items, uint32_t g) {
    const uint32_t n_units = (((n_items + 63u) >> 6) + 63u) >> 6;
    return n_units > g ? ((n_units - g + 7u) >> 3) * 16u : 0u;
}
__device__ __forceinline__ uint32_t queue_chunks(uint32_t n) { return (((n + 63u) >> 6) + 3u) >> 2; }

// (group << 28 | chunk) for iteration `it` of this workgroup, or CRT_NO_WORK; uniform over the workgroup.
// A workgroup is 4 waves (256 threads), 2 waves or a single wave: with fewer than 4 waves per workgroup, 4 / W
// consecutive workgroups of the same XCD stand for one chunk, so the dispatcher refills CUs (half-)wave-pair by wave.
struct WaveId { uint32_t lane, wave, lds_wave, vblock, vgrid, quadrant, sub; };
// four_per_batch (lane_samples): four consecutive single-wave workgroups of an XCD slice stand for ONE batch, one 4 x 4 pixel quadrant each
// sub_log2 (k_trace, single-wave workgroups): 1 << sub_log2 consecutive workgroups of an XCD slice stand for ONE wave of the mapping
// below, `sub` says which of them this is
__device__ __forceinline__ WaveId wave_id(bool one_batch_per_workgroup = false, bool four_per_batch = false, uint32_t sub_log2 = 0u) {
    WaveId w;
    w.lane = threadIdx.x & 63u;
    w.lds_wave = threadIdx.x >> 6;
    w.sub = 0u;
    if (four_per_batch) {
        const uint32_t qq = blockIdx.x >> 3;
        w.quadrant = qq & 3u;
        const uint32_t q = qq >> 2;
        w.wave = q & 3u;
        w.vblock = ((q >> 2) << 3) | (blockIdx.x & 7u);
        w.vgrid = gridDim.x >> 4;
        return w;
    }
    w.quadrant = 0u;
    // W = 1, 2 or 4 waves per workgroup: 4 / W consecutive workgroups of the same XCD slice stand for one 4-batch chunk.
    // one_batch_per_workgroup (wave_samples): the workgroup's waves all work on ONE batch, so it maps like a single wave.
    const uint32_t W = one_batch_per_workgroup ? 1u : blockDim.x >> 6, per_log2 = W == 1u ? 2u : W == 2u ? 1u : 0u;
    w.sub = (blockIdx.x >> 3) & ((1u << sub_log2) - 1u);
    const uint32_t q = blockIdx.x >> (3u + sub_log2);
    w.wave = (q & ((1u << per_log2) - 1u)) * W + (one_batch_per_workgroup ? 0u : w.lds_wave);
    w.vblock = ((q >> per_log2) << 3) | (blockIdx.x & 7u);
    w.vgrid = gridDim.x >> (per_log2 + sub_log2);
    return w;
}

template <bool DENSE>
__device__ __forceinline__ uint32_t static_chunk(const WaveId& w, const uint32_t* counts, uint32_t n_dense, uint32_t it) {
    const uint32_t g = w.vblock & 7u;
    const uint32_t nch = DENSE ? dense_chunks_of_group(n_dense, g) : queue_chunks(counts[g * CRT_COUNTER_STRIDE]);
    const uint32_t c = (w.vblock >> 3) + it * (w.vgrid >> 3);     // the host launches a multiple of 8 workgroups
    return c < nch ? (g << 28) | c : CRT_NO_WORK;
}

// Pool-based kernels: a workgroup chunk is 1024 items, 256 consecutive items per wave (its refill pool).
__device__ __forceinline__ uint32_t dense_pool_chunks_of_group(uint32_t n_items, uint32_t g) {
    const uint32_t n_units = (n_items + 4095u) >> 12;
    return n_units > g ? ((n_units - g + 7u) >> 3) * 4u : 0u;
}
__device__ __forceinline__ uint32_t queue_pool_chunks(uint32_t n) { return (n + 1023u) >> 10; }
template <bool DENSE>
__device__ __forceinline__ uint32_t static_pool_chunk(const WaveId& w, const uint32_t* counts, uint32_t n_dense, uint32_t it) {
    const uint32_t g = w.vblock & 7u;
    const uint32_t nch = DENSE ? dense_pool_chunks_of_group(n_dense, g) : queue_pool_chunks(counts[g * CRT_COUNTER_STRIDE]);
    const uint32_t c = (w.vblock >> 3) + it * (w.vgrid >> 3);
    return c < nch ? (g << 28) | c : CRT_NO_WORK;
}
// first item of this wave's pool for a dense pool chunk
__device__ __forceinline__ uint32_t dense_pool_first(uint32_t v, uint32_t wave) {
    const uint32_t g = v >> 28, c = v & 0x0fffffffu;
    const uint32_t unit = (c >> 2) * 8u + g;
    return unit * 4096u + (c & 3u) * 1024u + wave * 256u;
}

// index of this lane's item for a dense chunk (>= n_items when past the end)
__device__ __forceinline__ uint32_t dense_item(uint32_t v, uint32_t wave, uint32_t lane) {
    const uint32_t g = v >> 28, c = v & 0x0fffffffu;
    const uint32_t unit = (c >> 4) * 8u + g;
    return (unit * 64u + (c & 15u) * 4u + wave) * 64u + lane;
}

// totals[0], [1] += lane visits (nodes, triangles); totals[4], [5] += wave-level steps of the same blocks
__device__ __forceinline__ void flush_visit_totals(unsigned long long* totals, uint32_t nn, uint32_t nt, uint32_t wn = 0u, uint32_t wt = 0u) {
    unsigned long long a = nn, b = nt, c = wn, d = wt;
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
        c += __shfl_down(c, off);
        d += __shfl_down(d, off);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (a) atomicAdd(&totals[0], a);
        if (b) atomicAdd(&totals[1], b);
        if (c) atomicAdd(&totals[4], c);
        if (d) atomicAdd(&totals[5], d);
    }
}

// Trace kernel over an explicit ray buffer (crt_trace / crt_trace_device): pools of 256 >> a.pool_split_log2 rays per wave, lane refill.
template <bool ANY, bool STATS>
__global__ void __launch_bounds__(CRT_TRACE_BLOCK) k_trace(TraceArgs a) {
    extern __shared__ uint2 s_lds[];     // traversal stacks [wave][level][lane]
    const WaveId wid = wave_id(false, false, a.pool_split_log2);
    const uint32_t lane = wid.lane, wave = wid.wave;
    uint2* stk = s_lds + (size_t)wid.lds_wave * (a.stack_entries + CRT_HIT_SLOTS) * 64u + lane;
    const uint32_t n = a.count_ptr ? *a.count_ptr : a.n;
    uint32_t nn_total = 0, nt_total = 0;
    CRT_CHUNK_LOOP(it) {
        const uint32_t v = static_pool_chunk<true>(wid, nullptr, n, it);
        if (v == CRT_NO_WORK) break;
        const uint32_t pool = 256u >> a.pool_split_log2;
        const uint32_t first = dense_pool_first(v, wave) + wid.sub * pool;
        if (first >= n) continue;
        const uint32_t last = first + pool < n ? first + pool : n;
        uint32_t nn = 0, nt = 0, wn_unused = 0, wt_unused = 0;
        // per-ray counters need the count of ONE ray: sample the running totals at load and at completion
        uint32_t nn0 = 0, nt0 = 0;
        auto load = [&](uint32_t i, vec3& o, vec3& d, float& tmax) {
                const float4 r0 = a.rays[2 * (size_t)i], r1 = a.rays[2 * (size_t)i + 1];
                o = V3(r0.x, r0.y, r0.z); d = V3(r1.x, r1.y, r1.z); tmax = r0.w;
                nn0 = nn; nt0 = nt;
                return true;
            };
        auto done = [&](uint32_t i, const HitState& best, bool hit) {
                float4 h;
                h.x = ANY ? 0.f : (hit ? best.t : 0.f);
                h.y = ANY ? 0.f : best.u;
                h.z = ANY ? 0.f : best.v;
                int out = best.tri;
                if (ANY) out = hit ? 0 : -1;
                else if (a.out_orig_id) out = hit ? best.id : -1;
                h.w = __int_as_float(out);
                a.hits[i] = h;
                if (STATS) {
                    const uint32_t dn = nn - nn0, dt = nt - nt0;
                    a.stats[i] = ((dt > 65535u ? 65535u : dt) << 16) | (dn > 65535u ? 65535u : dn);
                }
            };
        walk_pool<ANY, STATS>(a.nodes, a.tris, stk - lane, (int)a.stack_entries, a.overflow, PoolStatic{first, last}, a.refill_min, a.tri_min, 0u, load, done, nn, nt, wn_unused, wt_unused);
        nn_total += nn; nt_total += nt;
    }
    (void)nn_total; (void)nt_total; (void)lane;
}

// ------------------------------------------------------------------ BVH2 (the reference's live path) ----

// path_trace.fs:84-109
__device__ __forceinline__ float hit_bbox2(vec3 o, vec3 bmin, vec3 bmax, vec3 invdir, float& tl) {
    bmin = (bmin - o) * invdir;
    bmax = (bmax - o) * invdir;
    const vec3 tmax = V3(__builtin_fmaxf(bmax.x, bmin.x), __builtin_fmaxf(bmax.y, bmin.y), __builtin_fmaxf(bmax.z, bmin.z));
    const vec3 tmin = V3(__builtin_fminf(bmax.x, bmin.x), __builtin_fminf(bmax.y, bmin.y), __builtin_fminf(bmax.z, bmin.z));
    const float th = __builtin_fminf(tmax.x, __builtin_fminf(tmax.y, tmax.z));
    tl = __builtin_fmaxf(tmin.x, __builtin_fmaxf(tmin.y, tmin.z));
    return th;
}

// The walk the reference actually ships: binary BVH, two-child slab test, near child first, far child on a
// per-ray stack (path_trace.fs:511-652 closest hit, :669-819 any hit), on the FlatNode array as uploaded
// (Scene.h:1057-1062).  Raw 1/d like the shader.  tie = 0 keeps the shader's first-visited rule (strict '<'),
// tie = 1 is the lowest-original-id rule of the CWBVH path.  Lock-step 64-ray batches, int stack in LDS.
// One ray through the BVH2.  best.tri = BVH2 leaf slot (index into the slot-ordered records), best.id = original id;
// returns hit / occluded.  `stk` is this lane's column of an int stack, stk[level * 64].
template <bool ANY, bool STATS>
__device__ __forceinline__ bool traverse_bvh2(const float4* __restrict__ nodes, const float4* __restrict__ tris, vec3 o, vec3 d,
                                              float tmax_in, uint32_t tie, int* stk, int stack_entries, uint32_t* overflow, HitState& best,
                                              uint32_t& nn, uint32_t& nt) {
    const vec3 invdir = V3(rcp_ieee(d.x), rcp_ieee(d.y), rcp_ieee(d.z));
    best.t = tmax_in; best.u = 0.f; best.v = 0.f; best.tri = -1; best.id = -1;
    int ptr = 0;
    stk[0] = -1; ptr = 1;
    int ind = 0;
    while (ind > -1) {
        const float4 bmin = nodes[2 * (size_t)ind], bmax = nodes[2 * (size_t)ind + 1];
        if (STATS) ++nn;
        const int left = link_of(bmin.w);            // an index as a float, or its bit pattern from 2^24 on (host/flatnode_link.hpp)
        if (bmax.w == 0.0f) {
            const float4 amin = nodes[2 * (size_t)left], amax = nodes[2 * (size_t)left + 1];
            const float4 cmin = nodes[2 * (size_t)left + 2], cmax = nodes[2 * (size_t)left + 3];
            float tl1, tl2;
            const float th1 = hit_bbox2(o, V3(amin.x, amin.y, amin.z), V3(amax.x, amax.y, amax.z), invdir, tl1);
            const float th2 = hit_bbox2(o, V3(cmin.x, cmin.y, cmin.z), V3(cmax.x, cmax.y, cmax.z), invdir, tl2);
            bool l, r;
            if (ANY) {                                   // path_trace.fs:741-742
                l = th1 >= 0 && th1 >= tl1 && tl1 <= best.t;
                r = th2 >= 0 && th2 >= tl2 && tl2 <= best.t;
            } else {                                     // path_trace.fs:562-563 ('<=' under the lowest-id rule)
                l = th1 > 0 && th1 >= tl1 && (tie ? tl1 <= best.t : tl1 < best.t);
                r = th2 > 0 && th2 >= tl2 && (tie ? tl2 <= best.t : tl2 < best.t);
            }
            if (l) {
                ind = left;
                if (r) {
                    const int off = tl1 > tl2 ? 1 : 0;
                    if (ptr < stack_entries) { stk[ptr * 64] = ind + 1 - off; ++ptr; } else atomicAdd(overflow, 1u);
                    ind += off;
                }
                continue;
            } else if (r) {
                ind = left + 1;
                continue;
            }
        } else {
            const int range = (int)bmax.w;
            for (int s = left; s < left + range; ++s) {
                const float4* tp = tris + 3 * (size_t)s;
                const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                if (STATS) ++nt;
                float u, vv, t;
                if (mt_test(ta, tb, tc, o, d, u, vv, t)) {
                    if (ANY) {
                        if (t < best.t) { best.tri = s; return true; }
                    } else {
                        const int id = __float_as_int(ta.w);
                        if (t < best.t || (tie && t == best.t && best.tri >= 0 && id < best.id)) {
                            best.t = t; best.u = u; best.v = vv; best.tri = s; best.id = id;
                        }
                    }
                }
            }
        }
        --ptr;
        ind = stk[ptr * 64];
    }
    return best.tri >= 0;
}

template <bool ANY, bool STATS>
__global__ void __launch_bounds__(CRT_TRACE_BLOCK) k_trace_bvh2(Bvh2Args a) {
    extern __shared__ int s_stk2[];      // [wave][level][lane]
    const WaveId wid = wave_id();
    const uint32_t lane = wid.lane, wave = wid.wave;
    int* stk = s_stk2 + (size_t)wid.lds_wave * a.stack_entries * 64u + lane;
    CRT_CHUNK_LOOP(it) {
        const uint32_t v = static_chunk<true>(wid, nullptr, a.n, it);
        if (v == CRT_NO_WORK) break;
        const uint32_t i = dense_item(v, wave, lane);
        if (i >= a.n) continue;
        const float4 r0 = a.rays[2 * (size_t)i], r1 = a.rays[2 * (size_t)i + 1];
        HitState best;
        uint32_t nn = 0, nt = 0;
        const bool hit = traverse_bvh2<ANY, STATS>(a.nodes, a.tris, V3(r0.x, r0.y, r0.z), V3(r1.x, r1.y, r1.z), r0.w, a.tie, stk,
                                                   (int)a.stack_entries, a.overflow, best, nn, nt);
        float4 h;
        h.x = ANY ? 0.f : (hit ? best.t : 0.f);
        h.y = ANY ? 0.f : best.u;
        h.z = ANY ? 0.f : best.v;
        h.w = __int_as_float(ANY ? (hit ? 0 : -1) : best.id);
        a.hits[i] = h;
        if (STATS) a.stats[i] = ((nt > 65535u ? 65535u : nt) << 16) | (nn > 65535u ? 65535u : nn);
    }
}

// ------------------------------------------------------------------ queue helpers ----

// Wave-aggregated append: one atomic per wave, lanes get consecutive slots in lane order.
__device__ __forceinline__ uint32_t wave_append(bool want, uint32_t* counter) {
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
    base = __shfl(base, leader);
    return base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
}

#include "rt_sort.hpp"      // regrouping rays between launches: bounce-ray bins (option ray_bins), the deferred shadow rays sorted by origin cell (sort_shadow)

// local pixel index -> frame pixel.  Pixels are laid out tile-major; inside a tile, 8x8 blocks
// row-major, so one 64-lane wave covers an 8x8 pixel block (coherent primary rays).
// A wave-uniform value passed through an empty asm statement: what is computed from it (the reciprocal of an integer divisor, an
// int -> float conversion) is computed where it is used instead of being hoisted to the top of the kernel, where it would hold a
// VGPR across every loop (wave-uniform floats and division constants live in VGPRs: there is no scalar float unit).
__device__ __forceinline__ uint32_t here(uint32_t uniform) { asm volatile("" : "+s"(uniform)); return uniform; }

__device__ __forceinline__ bool pixel_of(const FrameArgs& f, uint32_t i, uint32_t& px, uint32_t& py) {
    uint32_t t, bx, by;
    const uint32_t k = i & 63u;
    if (f.tile_log2) {                                   // power-of-two tiles (the default 64): no integer division
        const uint32_t s2 = 2u * f.tile_log2, sb = f.tile_log2 - 3u;
        t = i >> s2;
        const uint32_t blk = (i & ((1u << s2) - 1u)) >> 6;
        bx = blk & ((1u << sb) - 1u); by = blk >> sb;
    } else {
        const uint32_t tile_px = here(f.tile * f.tile);
        t = i / tile_px;
        const uint32_t blk = (i - t * tile_px) >> 6, blocks_per_row = here(f.tile >> 3);
        by = blk / blocks_per_row; bx = blk - by * blocks_per_row;
    }
    const uint2 txy = f.tile_xy[t];
    px = txy.x * f.tile + bx * 8u + (k & 7u);
    py = txy.y * f.tile + by * 8u + (k >> 3);
    return px < f.width && py < f.height;
}

// texture(albedo_textures, vec3(uv, layer)) — GL_LINEAR, GL_REPEAT, RGB8 (Scene.h:1065-1078); filter arithmetic as
// defined in oracle/oracle.c sample_albedo (texels were converted to c/255.0f at upload).
__device__ __forceinline__ int wrap_i(int i, int n) { const int m = i % n; return m < 0 ? m + n : m; }
__device__ __forceinline__ vec3 sample_albedo(const SegmentArgs& a, float u, float v, int layer) {
    const int W = a.tex_width, H = a.tex_height;
    layer = layer < 0 ? 0 : layer > a.n_textures - 1 ? a.n_textures - 1 : layer;
    const float* img = a.textures + (size_t)layer * (size_t)W * (size_t)H * 3;
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float x0 = __builtin_floorf(x), y0 = __builtin_floorf(y);
    float fx = x - x0, fy = y - y0;
    if (!(__builtin_fabsf(x0) < 1e9f)) { x0 = 0.f; fx = 0.f; }
    if (!(__builtin_fabsf(y0) < 1e9f)) { y0 = 0.f; fy = 0.f; }
    const int i0 = wrap_i((int)x0, W), i1 = wrap_i((int)x0 + 1, W);
    const int j0 = wrap_i((int)y0, H), j1 = wrap_i((int)y0 + 1, H);
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float t00 = img[3 * ((size_t)j0 * W + i0) + k], t10 = img[3 * ((size_t)j0 * W + i1) + k];
        const float t01 = img[3 * ((size_t)j1 * W + i0) + k], t11 = img[3 * ((size_t)j1 * W + i1) + k];
        const float top = t00 * (1.0f - fx) + t10 * fx;
        const float bot = t01 * (1.0f - fx) + t11 * fx;
        c[k] = top * (1.0f - fy) + bot * fy;
    }
    return V3(c[0], c[1], c[2]);
}

__device__ __forceinline__ void add_to_sum(float* __restrict__ sum, uint32_t pix, vec3 L) {   // path_trace.fs:1055-1059
    float* s = sum + 3 * (size_t)pix;
    s[0] = L.x + s[0];
    s[1] = L.y + s[1];
    s[2] = L.z + s[2];
}

// ------------------------------------------------------------------ materials ---------

// path_trace.fs:44-60
__device__ __forceinline__ void onb(vec3 n, vec3& ou, vec3& ov) {
    if (n.z < -0.9999999f) { ou = V3(0.f, -1.f, 0.f); ov = V3(-1.f, 0.f, 0.f); }
    else {
        const float aa = rcp_ieee(1.0f + n.z);
        const float bb = -n.x * n.y * aa;
        ou = V3(1.0f + bb, bb, -n.x);
        ov = V3(bb, 1.0f + bb, -n.y);
    }
}

// Mirror and the GGX / Disney-diffuse lobe: NO REFERENCE CODE exists for these (the reference has the MaterialType enum,
// `albedo.w = Mirror_type` in its loader and the shader's specular.w / is_specular plumbing, README.md:23 promises a Disney
// BSDF).  They are defined by oracle/oracle.c ("materials beyond Lambert"); every function below restates the oracle's
// operation for operation, so frames with such materials stay bit-identical to it.
#define CRT_MIRROR_TYPE 1.0f    // Scene.h:114
#define CRT_DISNEY_TYPE 17.0f   // Scene.h:131

struct Disney { vec3 base; float metallic, rough, a2, p_spec; };

__device__ __forceinline__ float schlick5(float c) {
    float m = 1.0f - c;
    if (m < 0.0f) m = 0.0f;
    if (m > 1.0f) m = 1.0f;
    const float m2 = m * m;
    return (m2 * m2) * m;
}
__device__ __forceinline__ float smith_g1(float c, float a2) { return __fdiv_rn(2.0f * c, c + sqrt_ieee(a2 + (1.0f - a2) * (c * c))); }

__device__ __forceinline__ Disney disney_params(vec3 base, float me, float ro) {
    Disney m;
    m.base = base;
    m.metallic = me < 0.0f ? 0.0f : me > 1.0f ? 1.0f : me;
    m.rough = ro < 0.03f ? 0.03f : ro > 1.0f ? 1.0f : ro;
    const float al = m.rough * m.rough;
    m.a2 = al * al;
    m.p_spec = 0.5f + 0.5f * m.metallic;
    return m;
}

// f(wo, wi) without the cosine and the solid-angle pdf of disney_sample; both 0 below the horizon
__device__ __forceinline__ void disney_eval(const Disney& m, vec3 n, vec3 wo, vec3 wi, vec3& f, float& pdf) {
    f = V3(0.f, 0.f, 0.f);
    pdf = 0.f;
    const float nl = dot(n, wi), nv = dot(n, wo);
    if (!(nl > 0.0f && nv > 0.0f)) return;
    const vec3 hs = wi + wo;
    const float hh = dot(hs, hs);
    if (!(hh > 0.0f)) return;
    const vec3 h = hs * rcp_ieee(sqrt_ieee(hh));
    const float nh = dot(n, h), lh = dot(wi, h);
    if (!(nh > 0.0f && lh > 0.0f)) return;
    const float fl = schlick5(nl), fv = schlick5(nv), fh = schlick5(lh);
    const float fd90m1 = (0.5f + (2.0f * (lh * lh)) * m.rough) - 1.0f;
    const float fd = (1.0f + fd90m1 * fl) * (1.0f + fd90m1 * fv);
    const float kd = __fdiv_rn((1.0f - m.metallic) * fd, CRT_PI);
    const float t = (nh * nh) * (m.a2 - 1.0f) + 1.0f;
    const float D = __fdiv_rn(m.a2, (CRT_PI * t) * t);
    const float G = smith_g1(nl, m.a2) * smith_g1(nv, m.a2);
    const float spec = __fdiv_rn(D * G, (4.0f * nl) * nv);
    const float dmm = 0.04f * (1.0f - m.metallic);
    const vec3 F0 = V3(dmm + m.base.x * m.metallic, dmm + m.base.y * m.metallic, dmm + m.base.z * m.metallic);
    const vec3 F = V3(F0.x + (1.0f - F0.x) * fh, F0.y + (1.0f - F0.y) * fh, F0.z + (1.0f - F0.z) * fh);
    f = m.base * kd + F * spec;
    const float pdf_d = __fdiv_rn(nl, CRT_PI);
    const float pdf_s = __fdiv_rn(D * nh, 4.0f * lh);
    pdf = m.p_spec * pdf_s + (1.0f - m.p_spec) * pdf_d;
}

__device__ __forceinline__ vec3 disney_sample(const Disney& m, vec3 n, vec3 wo, float u0, float u1, float u2) {
    vec3 ou, ov;
    onb(n, ou, ov);
    const float phi = CRT_PI2 * u2;
    if (u0 < m.p_spec) {                                 // GGX normal distribution -> half vector -> reflect wo
        const float c2 = __fdiv_rn(1.0f - u1, 1.0f + (m.a2 - 1.0f) * u1);
        float s2 = 1.0f - c2;
        if (s2 < 0.0f) s2 = 0.0f;
        const float ch = sqrt_ieee(c2), sh = sqrt_ieee(s2);
        const vec3 hl = V3(sh * pinned_cos(phi), sh * pinned_sin(phi), ch);
        const vec3 h = (ou * hl.x + ov * hl.y) + n * hl.z;
        const float vh = dot(wo, h);
        return h * (2.0f * vh) - wo;
    }
    const float r = sqrt_ieee(u1);                       // path_trace.fs:257-270
    const vec3 dl = V3(r * pinned_cos(phi), r * pinned_sin(phi), sqrt_ieee(1.0f - u1));
    return (ou * dl.x + ov * dl.y) + n * dl.z;
}

// ------------------------------------------------------------------ path segment -----
// The kernel's argument block through the kernarg segment pointer, passed through an empty asm statement: a member read through the
// result is a scalar load placed where it is written (the compiler cannot hoist it above the statement), not one of the loads the
// by-value argument gets in the kernel's entry block.
typedef const __attribute__((address_space(4))) SegmentArgs* KArgs;
__device__ __forceinline__ RayBins load_bins(KArgs k) {
    RayBins b;
    b.count = k->bins_out.count; b.cap = k->bins_out.cap; b.off = k->bins_out.off; b.ovf_count = k->bins_out.ovf_count;
    b.ovf_base = k->bins_out.ovf_base; b.per_lane = k->bins_out.per_lane;
    for (int i = 0; i < 3; ++i) { b.origin[i] = k->bins_out.origin[i]; b.scale[i] = k->bins_out.scale[i]; }
    return b;
}
__device__ __forceinline__ KArgs kernarg_here() {
    KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

#ifndef CRT_SEG_OCC
// Waves per SIMD the segment kernels are compiled for: 6 (80 VGPRs).  Until round 4 the bounce kernels needed 90-96 VGPRs and lost 20 % when
// forced to 80 (spills); since the register diet (hit record in LDS, path state fetched after the walk, shadow walk after the bounce
// emission) every hot kernel fits 80 without a spill, and the sixth wave is worth +3.7 % on four segments of the 1 M-triangle scene and
// +6 % on the Cornell box; a seventh (72 VGPRs) +0.8 % / -1.8 % (profiles/r04_occupancy_ab.txt).
#define CRT_SEG_OCC 6
#endif
#ifndef CRT_SEG_OCC_DEFERRED
// The bounce kernels that leave their shadow rays to k_shadow_deferred carry one walk, not two: they fit 64 VGPRs without a spill, and the eighth
// wave hides more of the L2-miss latency these segments wait on (issue_busy 0.48): four segments +1.5 %, two +1.7 %, Disney +1.7 % (profiles/r05_experiments.md)
#define CRT_SEG_OCC_DEFERRED 8
#endif
#ifndef CRT_SEG_OCC_FIRST
// The batched first-segment kernel <FIRST, INPLACE, BATCH, WIDE> (the headline's launch): 6 as well; round 3, same box, three runs each,
// 5 / 6 / 7 / 8 waves: 1 M triangles 1080p 12,205 / 12,384 / 12,032 / 10,875 Mray/s, 4K 13,980 / 14,267 / 13,979 / 12,759.
#define CRT_SEG_OCC_FIRST 6
#endif
#ifndef CRT_SEG_OCC_BATCH
// The other batched first-segment builds (the sample loop's state on top of everything else; materials) and the counting kernels: 5 waves, 96 VGPRs.  They run where a
// launch is only as long as its longest waves (a shard, a small frame: SegmentArgs::wide_first = 0), which finish sooner without scratch
// traffic (1/8 of a 1080p frame, 4 samples side by side: 0.0453 ms per frame at 5 waves, 0.0503 at 6), and on Mirror / Disney scenes.
#define CRT_SEG_OCC_BATCH 5
#endif

// One path segment per lane, fused: [ray generation (FIRST) | queue fetch] -> CWBVH closest hit -> shading (path_trace.fs:872-1018) ->
// emission of the next path ray with wave-ballot compaction -> the NEE shadow ray.  Nothing but the output queues (and, for paths that go
// on, 40 B of path state) touches HBM; a path that ends here adds its radiance to the sum buffer directly (or leaves it in l_final).
// TEX compiles the textured-albedo branch in (its double-precision pow costs registers the untextured path should not pay for); MAT the
// mirror / Disney branches.  Every hot build fits 80 VGPRs = 6 waves per SIMD (CRT_SEG_OCC*).
// Walks: walk_batch (the last rays of a draining wave get eight lanes each) walks the closest hits of every segment and the in-place shadow
// rays of the bounce segments.  The first segment's shadow rays are coherent and the voting loop costs them more than it gives (1 M
// triangles, 1080p: closest only 14,131 Mray/s, both 13,730, shadow only 13,618, neither 13,965 — profiles/r04_experiments.md): they keep
// the plain per-lane loop and only hand the wave's last eight rays to the group phase (traverse_any_then_groups).  a.tri_min == 0 (trees of
// a few nodes, e.g. the 32-triangle Cornell box: 0.0755 vs 0.0816 ms) or a.lanes_log2 == 0 selects the plain per-lane loops.
// PRETRACED (option bounce_refill): the closest hit of each queue entry was found by k_closest_queue (lane refill over pools of rays, which
// cannot be fused with lock-step shading); the kernel then only shades.
// INPLACE: the NEE shadow ray is walked right here (path_trace.fs:968, where the shader has it) — after the bounce ray has been sampled and
// queued, so that nothing but L, C and the path's index sits in registers through the walk.  The rays start where the closest-hit walk of
// the same lanes just ended, so their first nodes and triangles are still in L1, and no queue traffic, launch or second kernel's tail is paid
// (first segment, 1 M triangles: 0.214 + 0.158 ms as two kernels, 0.292 ms fused).
// !INPLACE (DEFERRED shadow rays, round 5): the ray goes to the frame's NEE queue with the index of a CONTRIBUTION SLOT (segment, path) that
// holds C, and k_shadow_deferred walks the shadow rays of all such segments in one full-occupancy launch after the last segment, clearing
// the slots of occluded rays.  The segment no longer carries L: what it adds to the path's radiance — C, or the emitter's T * e (:894-928) —
// sits in its slot, the path state keeps the mask of slots written, and k_fold_paths adds a path's slots in segment order afterwards: the
// additions the in-place form does, in the same order on the same operands.
// BVH2 (in place only): the closest-hit and the shadow walk are the shipped shader's own BVH2 walks (path_trace.fs:511-819, traverse_bvh2)
// on the FlatNode array — the live path of the reference as a frame renderer.
// BATCH (FIRST + INPLACE): a.n_samples samples per pixel in one launch (crt_render_frames), see the sample loop.  WIDE / ONE: see below.
template <bool FIRST, bool STATS, bool TEX, bool PRETRACED, bool INPLACE, bool BVH2 = false, bool MAT = false, bool BATCH = false, bool WIDE = false, bool ONE = false>
__global__ void __launch_bounds__(CRT_TRACE_BLOCK, ((WIDE || ONE) ? CRT_SEG_OCC_FIRST : (BATCH || STATS) ? CRT_SEG_OCC_BATCH : (!FIRST && !INPLACE && !PRETRACED) ? CRT_SEG_OCC_DEFERRED : CRT_SEG_OCC)) k_segment(SegmentArgs a) {
    extern __shared__ uint2 s_lds[];     // traversal stacks [wave][level][lane]
    // Uniform node steps are compiled into every first-segment kernel.  (In the single-sample kernel they lost while the uniform step still
    // converted bytes and the loops carried their flags — 8 x 8-pixel waves agree less than the 4 x 4-pixel waves of a batched launch, and the
    // node's SGPRs pushed loop state out of the scalar register file; with the float planes and the lean loops they win there too.)
    constexpr bool UNI_K = FIRST;
    // uniform: the workgroup's waves are the samples of one 64-pixel batch.  The 6-waves-per-SIMD build is never launched in that form
    // (launch_segment), and compiling the form out of it frees the registers its LDS result strip and wave index would hold
    const bool wave_samples = BATCH && !WIDE && !ONE && a.wave_samples == 1u;
    // uniform: a wave is one 4 x 4 pixel quadrant of a batch x 4 samples (lane = sample * 16 + pixel): the 64 rays of a wave leave a
    // quarter of the area, i.e. agree on their nodes like the rays of a frame of twice the resolution
    // ONE (with WIDE): the launch is four samples in the lanes form — one pass, known at compile time: around a sample loop whose trip count
    // is a run-time value the compiler hoists every constant and uniform condition of the body into SGPRs that then live across both walks
    // (and are spilled into VGPR lanes: 73 v_writelane + 86 v_readlane in the headline kernel before this)
    const bool lane_samples = BATCH && (ONE || a.wave_samples == 2u);
    const WaveId wid = wave_id(wave_samples, lane_samples);
    const uint32_t lane = wid.lane, wave = wid.wave;
    const uint32_t wave_stride = (a.stack_entries + CRT_HIT_SLOTS) * 64u;     // per-wave LDS region in uint2 units
    uint2* stk = s_lds + (size_t)wid.lds_wave * wave_stride + lane;
    int* stk2 = reinterpret_cast<int*>(s_lds) + (size_t)wid.lds_wave * a.stack_entries2 * 64u + lane;   // BVH2 mode
    const float4* const recs = BVH2 ? a.tris2 : a.tris;   // intersection records the hit index refers to
    const FrameArgs& f = a.f;
    uint32_t nn = 0, nt = 0, nn_any = 0, nt_any = 0;
    uint32_t wn = 0, wt = 0, wn_any = 0, wt_any = 0;     // wave-level step counts of the same blocks (counting kernels)
    uint32_t n_hits = 0;                                 // closest-hit rays that hit something, i.e. lanes that ran the shading code
    uint32_t nu = 0, nu_any = 0;                         // node visits that went through the scalar cache (uniform node steps)
    // Queue counters are double-banked by frame parity: the first kernel of a frame clears the bank the next
    // frame will append to (last touched by the previous frame, which stream order has retired).
    if (FIRST && a.zero_counts && blockIdx.x == 0)   // 64 or 256 threads, either works
        for (uint32_t i = threadIdx.x; i < a.n_zero; i += blockDim.x) a.zero_counts[i] = 0u;
    CRT_CHUNK_LOOP(it) {
        uint32_t v;
        if (!FIRST && a.bin_start) {
            // binned input: ONE queue in consumer order (bin after bin — octant-major — then the overflow entries); workgroup group
            // x (one XCD, as far as the round-robin dispatch goes) takes the x-th eighth of its chunks, i.e. about one direction
            // octant, so an L2 sees rays that head the same way
            const uint32_t n_chunks = queue_chunks(a.count_in[0]), per_group = (n_chunks + 7u) >> 3;
            const uint32_t grp = wid.vblock & 7u, c = (wid.vblock >> 3) + it * (wid.vgrid >> 3);
            v = (c < per_group && grp * per_group + c < n_chunks) ? (grp << 28) | (grp * per_group + c) : CRT_NO_WORK;
        } else {
            v = static_chunk<FIRST>(wid, a.count_in, f.n_local_pixels, it);
        }
        if (v == CRT_NO_WORK) break;
        const uint32_t g = v >> 28;                         // owner group of this chunk: its sub-queues get the output
        uint32_t* const count_shadow = a.count_shadow + g * CRT_COUNTER_STRIDE;
        uint32_t* const count_next = a.count_next + g * CRT_COUNTER_STRIDE;
        float4* const shadow_q = a.shadow + 2 * (size_t)g * a.sub_capacity;      // !INPLACE: this segment's region of the NEE queue
        float4* const next_q = a.rays_next + 2 * (size_t)g * a.sub_capacity;
        uint32_t e, n;
        uint32_t cost_tile = 0;
        bool cost_valid = false;
        uint32_t cost_t0 = 0;                               // low word of the cycle counter, wave-uniform: lives in an SGPR, not in a VGPR pair
        if (FIRST) {
            e = dense_item(v, wave, lane);
            if (lane_samples) {
                // pixel j = lane & 15 of quadrant q of the batch's 8 x 8 block (items of a batch are row-major in the block)
                const uint32_t j = lane & 15u, q = wid.quadrant;
                e = dense_item(v, wave, 0u) + (((q >> 1) * 4u + (j >> 2)) * 8u + (q & 1u) * 4u + (j & 3u));
            }
            n = f.n_local_pixels;
            // the unit of work says WHEN a tile is rendered, tile_order says WHICH tile that is: expensive tiles first, so that the
            // launch does not end on a few long waves (the same pixels, the same sums; 1 M triangles 0.356 -> 0.331 ms,
            // Cornell 0.082 -> 0.076 ms with nothing but a centre-out order).  A batch of 64 items never straddles two tiles.
            if (f.tile_order && e < n) {
                const uint32_t tile_px = f.tile * f.tile;
                const uint32_t slot = f.tile_log2 ? e >> (2u * f.tile_log2) : e / here(tile_px);
                cost_tile = f.tile_order[slot];
                e = cost_tile * tile_px + (e - slot * tile_px);
            }
            if (a.tile_cost) cost_t0 = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_readcyclecounter());
            // lane 0 reports the batch's cost for its tile (a batch never straddles two tiles): its tile and whether it has a pixel
            // at all, as wave-uniform scalars — a per-lane copy would sit in VGPRs across every loop of the kernel
            cost_tile = __builtin_amdgcn_readlane(cost_tile, 0);
            cost_valid = __builtin_amdgcn_readlane((uint32_t)(e < n), 0) != 0u;
        } else {
            e = ((v & 0x0fffffffu) * 4u + wave) * 64u + lane;
            n = a.bin_start ? a.count_in[0] : a.count_in[g * CRT_COUNTER_STRIDE];
        }
        // BATCH (crt_render_frames on a one-segment path): the lane renders a.n_samples samples of its pixel one after the
        // other — exactly what the same number of launches would do to this pixel, without their launch gaps and kernel
        // tails (1 M triangles, 4 samples: 0.299 -> 0.276 ms per frame; Cornell 0.080 -> 0.064).  A separate instantiation:
        // the loop-carried state costs the single-sample kernel 50 bytes of scratch per lane otherwise.
        const uint32_t ws_waves = blockDim.x >> 6;           // wave_samples: wave w renders samples w, w + W, w + 2 W, ...
        const uint32_t n_smp = (ONE || !BATCH) ? 1u : (wave_samples ? (a.n_samples + ws_waves - 1u) / ws_waves : lane_samples ? a.n_samples >> 2 : a.n_samples);
        const uint32_t e_of_chunk = e;
        for (uint32_t smp_it = 0; smp_it < n_smp; ++smp_it) {
        // BATCH: the pixel index goes through an empty asm statement at the top of every sample, so that what is derived from it
        // (pixel coordinates, the camera-space direction before jitter, tile addresses) is recomputed per sample — a few dozen
        // instructions — instead of being hoisted out of the sample loop and kept in registers across both traversal loops
        if (BATCH) { e = e_of_chunk; asm volatile("" : "+v"(e)); }
        const uint32_t smp = wave_samples ? smp_it * ws_waves + wid.lds_wave : lane_samples ? smp_it * 4u + (lane >> 4) : smp_it;
        float rv = BATCH ? a.rv_s[smp & 7u] : f.rv;
        if (lane_samples) {                                  // per-lane sample: a select chain, not an indexed read of the argument block
            rv = a.rv_s[0];
            for (uint32_t k = 1; k < 8u; ++k) rv = smp == k ? a.rv_s[k] : rv;
        }
        bool active = e < n && (!wave_samples || smp < a.n_samples);
        uint32_t pix = 0;
        vec3 o = V3(0.f, 0.f, 0.f), d = V3(0.f, 0.f, 1.f);
        vec3 L = V3(0.f, 0.f, 0.f), T = V3(1.f, 1.f, 1.f);
        float prev_pdf = 1.0f, sx = 0.f, sy = 0.f;
        bool is_specular = true, true_area = false;
        uint32_t slot_mask = 0u;                            // !INPLACE: contribution slots this path has written (bits 8..)
        if (FIRST) {                                        // path_trace.fs:1026-1047
            uint32_t px = 0, py = 0;
            pix = BATCH ? smp * f.n_local_pixels + e : e;     // the path's id: its pixel, or (sample, pixel) when a launch renders several samples
            active = active && pixel_of(f, e, px, py);
            sx = (float)px + 0.5f; sy = (float)py + 0.5f;
            const float W = (float)here(f.width), H = (float)here(f.height);
            float jx = 0.f, jy = 0.f;
            if (f.jitter) {
                const float r1 = 2.0f * shader_rand(sx, sy, rv);
                const float r2 = 2.0f * shader_rand(sx, sy, rv);
                jx = r1 < 1.0f ? sqrt_ieee(r1) - 1.0f : 1.0f - sqrt_ieee(2.0f - r1);
                jy = r2 < 1.0f ? sqrt_ieee(r2) - 1.0f : 1.0f - sqrt_ieee(2.0f - r2);
                jx = __fdiv_rn(jx, W * 0.5f);
                jy = __fdiv_rn(jy, H * 0.5f);
            }
            const float tx = __fdiv_rn((float)px + 0.5f, W), ty = __fdiv_rn((float)py + 0.5f, H);
            float dx = (2.0f * tx - 1.0f) + jx;
            float dy = (2.0f * ty - 1.0f) + jy;
            dx = dx * f.aspect_tan;                         // (W / H * tan(fov/2)), formed on the host like the oracle does
            dy = dy * f.tan_fov;
            const vec3 right = V3(f.cam_right[0], f.cam_right[1], f.cam_right[2]);
            const vec3 up = V3(f.cam_up[0], f.cam_up[1], f.cam_up[2]);
            const vec3 fwd = V3(f.cam_forward[0], f.cam_forward[1], f.cam_forward[2]);
            d = normalize((right * dx + up * dy) + fwd);
            o = V3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
        } else if (active) {
            const float4* rq;
            if (a.bin_start) {
                CRT_MARK("loop_begin bins");
                rq = a.rays_in + 2 * (size_t)binned_entry(a.bin_start, a.bin_off_in, a.ovf_base_in, e);
                CRT_MARK("loop_end");
            } else {
                rq = a.rays_in + 2 * ((size_t)g * a.sub_capacity + e);
            }
            const float4 r0 = rq[0], r1 = rq[1];
            o = V3(r0.x, r0.y, r0.z); d = V3(r1.x, r1.y, r1.z);
            pix = __float_as_uint(r1.w);                   // the path's state is fetched AFTER the walk (below): nothing of it is needed inside
        }

        HitState hit;
        hit.tri = -1;
        if (PRETRACED) {
            if (active) {
                const float4 h = a.hits_in[(size_t)g * a.sub_capacity + e];
                hit.t = h.x; hit.u = h.y; hit.v = h.z; hit.tri = __float_as_int(h.w);
            }
        } else if (BVH2) {
            if (active) traverse_bvh2<false, STATS>(a.nodes2, a.tris2, o, d, CRT_INF, a.tie, stk2, (int)a.stack_entries2, a.overflow, hit, nn, nt);
        } else if (!ONE && a.tri_min == 0u) {      // (a one-pass build is launched with the voting loop and the group phase on: launch_segment)
            if (active) traverse<false, STATS, UNI_K>(a.nodes, a.tris, o, d, CRT_INF, stk, (int)a.stack_entries, a.overflow, hit, nn, nt, wn, wt, &nu);
        } else {
            // lock-step batch (one ray per lane, no refill) through the voting traversal loop: lanes that have no
            // ray say so and finish immediately with no visits (1 M triangles: 0.397 -> 0.310 ms)
            // (primary rays all start at the camera: the origin of the first segment's walk stays in scalar registers, UNIFORM_O)
            // the last rays of the batch get eight lanes each (a.lanes_log2 = 0: never — then this is the lock-step voting loop alone)
            walk_batch<false, STATS, FIRST, UNI_K>(a.nodes, a.tris, stk - lane, (int)a.stack_entries, a.overflow, active, o, d, CRT_INF, a.tri_min, a.lanes_log2, hit,
                                                   nn, nt, wn, wt, V3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]), &nu, a.planes);
        }

        // Everything the shading code reads from the argument block is fetched HERE, behind the closest-hit walk, through a laundered
        // kernarg pointer: arguments used by value are all loaded in the kernel's first block and then live (or are spilled into VGPR lanes)
        // across both walks — ~40 scalar registers of pointers and counts the loops never look at.
        const KArgs ka = kernarg_here();
        if (!FIRST && active) {
            // Path state of a ray that came through the queue: radiance so far, throughput, RNG state, flags.  Fetched here, behind the
            // closest-hit walk, instead of where the ray is fetched: a dozen values the walk never looks at would otherwise sit in
            // VGPRs through its whole loop (the bounce kernels are the ones short of registers: 90-96 VGPRs at 5 waves per SIMD).
            asm volatile("" : "+v"(pix));                  // keeps the loads below the loop (they depend on this copy of the index)
            if (ka->l_final) {                               // a batched frame: the path belongs to sample pix / n_local_pixels, with that frame's randomVector
                const uint32_t smp_of = pix / f.n_local_pixels;
                rv = ka->rv_s[0];
                for (uint32_t k = 1; k < 8u; ++k) rv = smp_of == k ? ka->rv_s[k] : rv;
            }
            const float4 Tp = ka->pb.T[pix];
            const float2 sd = ka->pb.seed[pix];
            if (INPLACE) { const float4 Lp = ka->pb.L[pix]; L = V3(Lp.x, Lp.y, Lp.z); prev_pdf = Lp.w; }
            else prev_pdf = ka->pb.L[pix].w;                 // a deferred segment adds nothing to L itself: the radiance so far is fetched when the path ends
            T = V3(Tp.x, Tp.y, Tp.z);
            is_specular = (__float_as_uint(Tp.w) & 1u) != 0u;
            true_area = (__float_as_uint(Tp.w) & 2u) != 0u;
            slot_mask = __float_as_uint(Tp.w) & 0xffffff00u;
            sx = sd.x; sy = sd.y;
        }
        // !INPLACE: what this segment adds to the path's radiance goes to its contribution slot instead of into L (k_fold_paths adds the slots in
        // segment order afterwards); slot_mask = the slots written so far, bit 8 + segment, kept in the path state's flag word
        auto contribute = [&](vec3 c) {
            if (INPLACE) L = L + c;
            else { ka->contrib[pix] = make_float4(c.x, c.y, c.z, 1.0f); slot_mask |= ka->slot_bit; }
        };
        bool emit_shadow = false, emit_next = false, finished = active, pending = false;
        float pend_pdf = 0.f;
        float4 sh0 = make_float4(0, 0, 0, 0), sh1 = sh0, sh2 = sh0, nx0 = sh0, nx1 = sh0;
        if (active && hit.tri >= 0) {
            CRT_MARK("shade_begin");
            if (STATS) ++n_hits;
            const float t = hit.t, bu = hit.u, bv = hit.v;
            const size_t rec_rows = BVH2 ? 3u : (size_t)CRT_TRI_ROWS;            // the BVH2 walk's slot-ordered records stay packed
            const float4 tb = recs[rec_rows * (size_t)hit.tri + 1], tc = recs[rec_rows * (size_t)hit.tri + 2];
            const int slot = __float_as_int(tb.w), mtl = __float_as_int(tc.w);
            const int4 vn = ka->triangles[3 * (size_t)slot + 1];                    // path_trace.fs:440-454
            vec3 n;
            if (vn.w == 0) n = V3((float)vn.x, (float)vn.y, (float)vn.z);
            else {
                const float* N = ka->normals;
                const vec3 na = V3(N[3 * (size_t)vn.x], N[3 * (size_t)vn.x + 1], N[3 * (size_t)vn.x + 2]);
                const vec3 nb = V3(N[3 * (size_t)vn.y], N[3 * (size_t)vn.y + 1], N[3 * (size_t)vn.y + 2]);
                const vec3 nc = V3(N[3 * (size_t)vn.z], N[3 * (size_t)vn.z + 1], N[3 * (size_t)vn.z + 2]);
                const float w = 1.0f - bu - bv;
                n = (na * w + nb * bu) + nc * bv;
            }
            const float4 m_albedo = ka->materials[4 * (size_t)mtl], m_emission = ka->materials[4 * (size_t)mtl + 1],
                         m_specular = ka->materials[4 * (size_t)mtl + 2];
            const float cos_incident = dot(d, n);
            const vec3 original_n = n;
            if (cos_incident > 0) n = -n;
            if (m_emission.w != -1.0f) {                                          // path_trace.fs:894-928
                const vec3 em = V3(m_emission.x, m_emission.y, m_emission.z);
                if (is_specular) contribute(T * em);
                else {
                    vec3 ld = d * t;
                    const float len = length(ld);
                    ld = normalize(ld);
                    const float cos_light = -1.0f * dot(ld, n);
                    const float len2 = len * len;
                    const int li = (int)m_emission.w;
                    const float* ap = ka->lights + 18 * (size_t)li + 15;
                    float pdf_light = __fdiv_rn(len2, ap[0] * cos_light) * ap[1];
                    if (MAT && true_area) pdf_light = 2.0f * pdf_light;           // the previous vertex was a Disney one (below)
                    const float tt = prev_pdf * prev_pdf;                         // power_heuristic :214-218
                    const float w = __fdiv_rn(tt, pdf_light * pdf_light + tt);
                    contribute((T * em) * w);
                }
            } else {
                const vec3 hit_point = (o + d * t) + n * 0.0002f;                 // path_trace.fs:930
                vec3 albedo = V3(m_albedo.x, m_albedo.y, m_albedo.z);
                if (TEX) {                                                        // path_trace.fs:471-483
                    const float tex = ka->materials[4 * (size_t)mtl + 3].x;
                    if (tex != -1.0f && ka->textures != nullptr) {
                        const int4 vt = ka->triangles[3 * (size_t)slot + 2];
                        const float2 ta = ka->texcoords[vt.x], tb2 = ka->texcoords[vt.y], tc2 = ka->texcoords[vt.z];
                        const float w = 1.0f - bu - bv;
                        const float tu = (ta.x * w + tb2.x * bu) + tc2.x * bv;
                        const float tv = (ta.y * w + tb2.y * bu) + tc2.y * bv;
                        const vec3 c = sample_albedo(a, tu, tv, (int)tex);
                        albedo = V3((float)pow((double)c.x, (double)2.2f), (float)pow((double)c.y, (double)2.2f),
                                    (float)pow((double)c.z, (double)2.2f));
                    }
                }
                // Materials beyond Lambert (MAT scenes only; oracle-defined, oracle.c "materials beyond Lambert" — the reference
                // has the hooks but no code): albedo.w = MaterialType, 1 = Mirror_type (Scene.h:114, :576-582), 17 = Disney_type (:131)
                const float mtype = m_albedo.w;
                if (MAT && mtype == CRT_MIRROR_TYPE) {
                    // perfect reflection: no NEE, no RNG draws, the next emitter hit takes the is_specular branch (:896)
                    const vec3 ns = normalize(n);
                    const float dn = dot(d, ns);
                    T = T * albedo;
                    if (!ka->last_segment) {
                        const vec3 rdir = d - ns * (2.0f * dn);
                        if (INPLACE || FIRST) ka->pb.L[pix] = make_float4(L.x, L.y, L.z, prev_pdf);      // (a deferred bounce segment leaves (L, pdf) as they are)
                        ka->pb.T[pix] = make_float4(T.x, T.y, T.z, __uint_as_float(1u | slot_mask));
                        ka->pb.seed[pix] = make_float2(sx, sy);
                        emit_next = true;
                        finished = false;
                        nx0 = make_float4(hit_point.x, hit_point.y, hit_point.z, CRT_INF);
                        nx1 = make_float4(rdir.x, rdir.y, rdir.z, __uint_as_float(pix));
                    }
                } else {
                    const bool disney = MAT && mtype == CRT_DISNEY_TYPE;
                    vec3 ns = n;
                    const vec3 wo = -d;
                    Disney dm;
                    if (disney) { ns = normalize(n); dm = disney_params(albedo, m_specular.x, m_specular.y); }
                    if (m_specular.w == 0.0f) {
                        if (ka->n_lights <= 0) {
                            shader_rand(sx, sy, rv); shader_rand(sx, sy, rv); shader_rand(sx, sy, rv);
                        } else {
                            int li = (int)(shader_rand(sx, sy, rv) * (float)(int)here((uint32_t)ka->n_lights));
                            if (li > ka->n_lights - 1) li = ka->n_lights - 1;
                            const float* Lt = ka->lights + 18 * (size_t)li;
                            const float sq = sqrt_ieee(shader_rand(sx, sy, rv));    // :843-855
                            const float b0 = 1.0f - sq;
                            const float b1 = shader_rand(sx, sy, rv) * sq;
                            const vec3 lp = (V3(Lt[0], Lt[1], Lt[2]) + V3(Lt[3], Lt[4], Lt[5]) * b0) + V3(Lt[6], Lt[7], Lt[8]) * b1;
                            vec3 ldir = lp - hit_point;
                            const float len = length(ldir);
                            const float ilen = rcp_ieee(len);
                            ldir = ldir * ilen;
                            const float cos_mtl = dot(ldir, original_n);
                            const float cos_light = dot(ldir, V3(Lt[9], Lt[10], Lt[11]));
                            bool lit = cos_mtl > 0.0f && cos_light < 0.0f;            // :968 (the occlusion test follows: in place, or deferred to k_shadow_deferred)
                            if (disney) lit = lit && dot(ns, ldir) > 0.0f;            // the lobe is zero below the shading horizon: no ray
                            if (lit) {
                                const vec3 le = V3(Lt[12], Lt[13], Lt[14]);
                                float pdf_light = __fdiv_rn(len * len, Lt[15] * -cos_light) * Lt[16];
                                vec3 c;
                                if (disney) {
                                    pdf_light = 2.0f * pdf_light;                     // true triangle area: Light.area is |u x v| (Scene.h:865-875)
                                    vec3 fr; float pdf_b;
                                    disney_eval(dm, ns, wo, ldir, fr, pdf_b);
                                    const float tt = pdf_light * pdf_light;
                                    const float w = __fdiv_rn(tt, pdf_b * pdf_b + tt);
                                    c = ((T * le) * fr) * (dot(ns, ldir) * w);
                                } else {
                                    const float bsdf_pdf = __fdiv_rn(dot(ldir, n) * 1.0f, CRT_PI);
                                    const float tt = pdf_light * pdf_light;
                                    const float w = __fdiv_rn(tt, bsdf_pdf * bsdf_pdf + tt);
                                    c = ((T * le) * albedo) * w;
                                }
                                c = V3(__fdiv_rn(c.x, pdf_light), __fdiv_rn(c.y, pdf_light), __fdiv_rn(c.z, pdf_light));
                                if (INPLACE) {
                                    // the occlusion test of path_trace.fs:968 is walked in this kernel — below, after the bounce ray has
                                    // been sampled and queued: nothing but L, C and the path's index then sits in registers through the
                                    // walk's loop (the shader walks it right here; the order of two independent computations is all that changes)
                                    const unsigned long long m = __ballot(true);
                                    if ((int)lane == __builtin_ctzll(m)) atomicAdd(count_shadow, (uint32_t)__builtin_popcountll(m));   // ray count only
                                    pending = true;
                                    sh2 = make_float4(c.x, c.y, c.z, 0.f);
                                } else {
                                    // deferred: C waits in this segment's slot of the path, the ray goes to the NEE queue (below)
                                    emit_shadow = true;
                                    contribute(c);
                                }
                                sh0 = make_float4(hit_point.x, hit_point.y, hit_point.z, len - CRT_EPS);
                                sh1 = make_float4(ldir.x, ldir.y, ldir.z, __uint_as_float(INPLACE ? pix : ka->slot_first + pix));
                            }
                        }
                    }
                    if (!ka->last_segment) {
                        vec3 sdir;
                        float bsdf_pdf;
                        bool go_on = true;
                        if (disney) {
                            const float u0 = shader_rand(sx, sy, rv);
                            const float u1 = shader_rand(sx, sy, rv);
                            const float u2 = shader_rand(sx, sy, rv);
                            sdir = disney_sample(dm, ns, wo, u0, u1, u2);
                            vec3 fr;
                            disney_eval(dm, ns, wo, sdir, fr, bsdf_pdf);
                            go_on = bsdf_pdf > 0.0f;                                 // sampled below the horizon: the path ends
                            if (go_on) T = T * (fr * __fdiv_rn(dot(ns, sdir), bsdf_pdf));
                        } else {
                            vec3 ou, ov;                                              // path_trace.fs:44-60
                            onb(n, ou, ov);
                            const float u1 = shader_rand(sx, sy, rv);               // :257-270
                            const float u2 = shader_rand(sx, sy, rv);
                            const float r = sqrt_ieee(u1);
                            const float phi = CRT_PI2 * u2;
                            const vec3 dl = V3(r * pinned_cos(phi), r * pinned_sin(phi), sqrt_ieee(1.0f - u1));
                            sdir = (ou * dl.x + ov * dl.y) + n * dl.z;
                            bsdf_pdf = __fdiv_rn(dot(sdir, n) * 1.0f, CRT_PI);
                            T = T * albedo;
                        }
                        if (go_on) {
                            if (INPLACE && pending) pend_pdf = bsdf_pdf;              // whoever walks the shadow ray writes L (+ C) and this pdf
                            else if (INPLACE || FIRST) ka->pb.L[pix] = make_float4(L.x, L.y, L.z, bsdf_pdf);
                            else reinterpret_cast<float*>(ka->pb.L + pix)[3] = bsdf_pdf;      // deferred bounce segment: the radiance so far stays where it is
                            ka->pb.T[pix] = make_float4(T.x, T.y, T.z, __uint_as_float((disney ? 2u : 0u) | slot_mask));   // bit 0 is_specular, bit 1 true_area, bits 8.. slots written
                            ka->pb.seed[pix] = make_float2(sx, sy);
                            emit_next = true;
                            finished = false;
                            nx0 = make_float4(hit_point.x, hit_point.y, hit_point.z, CRT_INF);
                            nx1 = make_float4(sdir.x, sdir.y, sdir.z, __uint_as_float(pix));
                        }
                    }
                }
            }
        }
        if (ka->bins_out.count) {
            const RayBins bins_out = load_bins(ka);
            CRT_MARK("loop_begin bins");      // the optional bins are not part of the instruction model (tools/roofline.py): bracketed like a loop
            // the wave's own stack region is free between the walks: the LDS table of the ranked append (3 KB) lives there when it fits
            uint32_t* const tab = wave_stride * 8u >= 3072u ? reinterpret_cast<uint32_t*>(s_lds + (size_t)wid.lds_wave * wave_stride) : nullptr;
            const uint32_t ni = bin_append(bins_out, emit_next, ray_bin_key(bins_out, nx0, nx1), tab);
            if (emit_next) { ka->rays_next[2 * (size_t)ni] = nx0; ka->rays_next[2 * (size_t)ni + 1] = nx1; }
            CRT_MARK("loop_end");
        } else {
            const uint32_t ni = wave_append(emit_next, count_next);
            if (emit_next) { next_q[2 * (size_t)ni] = nx0; next_q[2 * (size_t)ni + 1] = nx1; }
        }
        if (INPLACE) {
            // ---- the NEE shadow rays of this wave, walked now that the next segment's ray is out of the registers ----
            const bool lanes_walks = !BVH2 && (ONE || (a.lanes_log2 != 0u && a.tri_min != 0u));
            if (FIRST && lanes_walks) {
                // the first segment's shadow rays: the plain loop, then groups for the last rays of the wave
                const bool occluded = traverse_any_then_groups<STATS, UNI_K>(a.nodes, a.tris, stk - lane, (int)a.stack_entries, a.overflow, pending, V3(sh0.x, sh0.y, sh0.z),
                                                                      V3(sh1.x, sh1.y, sh1.z), sh0.w, a.tri_min, a.lanes_log2, nn_any, nt_any, wn_any, wt_any, &nu_any, a.planes);
                if (pending && !occluded) L = L + V3(sh2.x, sh2.y, sh2.z);
            } else if (lanes_walks) {
                // lanes per ray grow as the wave's shadow rays drain (walk_batch)
                HitState shh;
                walk_batch<true, STATS, false, UNI_K>(a.nodes, a.tris, stk - lane, (int)a.stack_entries, a.overflow, pending, V3(sh0.x, sh0.y, sh0.z), V3(sh1.x, sh1.y, sh1.z),
                                               sh0.w, a.tri_min, a.lanes_log2, shh, nn_any, nt_any, wn_any, wt_any, V3(0.f, 0.f, 0.f), &nu_any, a.planes);
                if (pending && shh.tri < 0) L = L + V3(sh2.x, sh2.y, sh2.z);
            } else if (pending) {
                HitState sh;
                const vec3 so = V3(sh0.x, sh0.y, sh0.z), sd = V3(sh1.x, sh1.y, sh1.z);
                const bool occluded = BVH2
                    ? traverse_bvh2<true, STATS>(a.nodes2, a.tris2, so, sd, sh0.w, a.tie, stk2, (int)a.stack_entries2, a.overflow, sh, nn_any, nt_any)
                    : traverse<true, STATS, UNI_K>(a.nodes, a.tris, so, sd, sh0.w, stk, (int)a.stack_entries, a.overflow, sh, nn_any, nt_any, wn_any, wt_any, &nu_any);
                if (!occluded) L = L + V3(sh2.x, sh2.y, sh2.z);
            }
            if (pending && emit_next) a.pb.L[pix] = make_float4(L.x, L.y, L.z, pend_pdf);     // the path goes on: its radiance so far waits in the path state
            pending = false;                                                                   // a path that ended here adds L below
        } else {
            // ---- deferred: the shadow ray waits in this segment's region of the NEE queue for k_shadow_deferred ----
            const uint32_t si = wave_append(emit_shadow, count_shadow);
            if (emit_shadow) { shadow_q[2 * (size_t)si] = sh0; shadow_q[2 * (size_t)si + 1] = sh1; }
            // a path that ends in a deferred segment leaves the radiance it had gathered before (in-place segments: the path state's L) and
            // the mask of its slots; k_fold_paths adds the slots in segment order
            if (finished) {
                vec3 Lp = V3(0.f, 0.f, 0.f);
                if (!FIRST) { const float4 l = a.pb.L[pix]; Lp = V3(l.x, l.y, l.z); }
                a.l_final[pix] = make_float4(Lp.x, Lp.y, Lp.z, __uint_as_float(0x80000000u | (slot_mask >> 8)));
            }
            finished = false;
        }
        const KArgs kb = kernarg_here();                   // what is read from here on is fetched now, not carried through the walk above
        // a path that ends here with nothing pending adds its radiance to the running sum now
        if (finished && !pending) {
            // several samples per launch on a path of several segments: the samples of a pixel finish in different launches, so each
            // leaves its radiance at its own place and k_fold_paths adds them in the order the frames would have come
            if (kb->l_final) kb->l_final[pix] = make_float4(L.x, L.y, L.z, 0.f);
            else if (!wave_samples && !lane_samples && (L.x != 0.f || L.y != 0.f || L.z != 0.f)) add_to_sum(kb->sum, (FIRST && BATCH) ? e : pix, L);
        }
        if (lane_samples && !kb->l_final) {
            // the four samples of a pixel sit in lanes j, j + 16, j + 32, j + 48: lane j adds them in sample order — what the frames one
            // after the other would add, zero radiance skipped as everywhere
            const bool mine = finished && !pending;
            const float mx = mine ? L.x : 0.f, my = mine ? L.y : 0.f, mz = mine ? L.z : 0.f;
            // one read-modify-write of the pixel's sum for the four samples: s = L_k + s in sample order, in registers — the additions
            // add_to_sum would do one frame after the other, without three of the four loads, stores and address computations
            float rx[4], ry[4], rz[4];
            bool any_nz = false;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const int src = (int)((lane & 15u) + 16u * k);
                rx[k] = __shfl(mx, src); ry[k] = __shfl(my, src); rz[k] = __shfl(mz, src);
                any_nz = any_nz || rx[k] != 0.f || ry[k] != 0.f || rz[k] != 0.f;
            }
            if (lane < 16u && e < n && any_nz) {
                float* const sp3 = kb->sum + 3 * (size_t)e;
                float s0 = sp3[0], s1 = sp3[1], s2 = sp3[2];
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k)
                    if (rx[k] != 0.f || ry[k] != 0.f || rz[k] != 0.f) { s0 = rx[k] + s0; s1 = ry[k] + s1; s2 = rz[k] + s2; }
                sp3[0] = s0; sp3[1] = s1; sp3[2] = s2;
            }
        }
        if (wave_samples && !kb->l_final) {
            // the waves' samples of this batch, added in sample order by wave 0 (what the frames one after the other would add)
            // behind the workgroup's stacks, whose size per wave is the larger of the two kinds of stack as in launch_segment
            const uint32_t stack_words = BVH2 && a.stack_entries2 * 64u > 2u * wave_stride ? a.stack_entries2 * 64u : 2u * wave_stride;
            float4* const s_res = reinterpret_cast<float4*>(reinterpret_cast<uint32_t*>(s_lds) + (size_t)ws_waves * stack_words);
            const bool mine = finished && !pending;
            s_res[wid.lds_wave * 64u + lane] = mine ? make_float4(L.x, L.y, L.z, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
            if (wid.lds_wave == 0u && e < n) {
                for (uint32_t k = 0; k < ws_waves && smp_it * ws_waves + k < kb->n_samples; ++k) {
                    const float4 r = s_res[k * 64u + lane];
                    if (r.w != 0.f && (r.x != 0.f || r.y != 0.f || r.z != 0.f)) add_to_sum(kb->sum, e, V3(r.x, r.y, r.z));
                }
            }
            __syncthreads();                                       // persistent grids: the next pass reuses the strip
        }
        }   // samples
        if (FIRST && a.tile_cost && lane == 0u && cost_valid)
            atomicAdd(a.tile_cost + cost_tile, (uint32_t)__builtin_readcyclecounter() - cost_t0);     // a wave lives far less than 2^32 cycles
    }
    if (STATS && !PRETRACED) flush_visit_totals(a.visit_totals, nn, nt, wn, wt);
    if (STATS && INPLACE) flush_visit_totals(a.visit_totals + 2, nn_any, nt_any, wn_any, wt_any);     // deferred shadow rays: k_shadow_deferred counts them
    if (STATS) flush_visit_totals(a.visit_totals + 8, n_hits, 0u);
    if (STATS) flush_visit_totals(a.visit_totals + 10, nu, nu_any);
    (void)stk; (void)stk2;
}

// Closest hit for a device-written path-ray queue (segments >= 1; option bounce_refill): lane refill (walk_pool: a finished lane takes the
// next ray, the last eight rays get eight lanes each); hits go to a buffer parallel to the queue and k_segment<PRETRACED> shades them.
// a.persistent: a grid of as many single-wave workgroups as the chip holds waves, every wave drawing rays from the eight sub-queues through
// their cursors until all are dry (PoolStream); else one workgroup per pool of a.pool rays of a sub-queue (PoolStatic).
template <bool STATS>
__global__ void __launch_bounds__(64, CRT_SEG_OCC) k_closest_queue(QueueTraceArgs a) {
    extern __shared__ uint2 s_lds[];     // this wave's traversal stack [level][lane] + hit slots
    const uint32_t g = blockIdx.x & 7u, c = blockIdx.x >> 3;
    uint32_t nn = 0, nt = 0, wn = 0, wt = 0;
    auto load = [&](uint32_t e, vec3& o, vec3& d, float& tmax) {        // e: flat entry, sub-queue * sub_capacity + place
        const float4 r0 = a.rays[2 * (size_t)e], r1 = a.rays[2 * (size_t)e + 1];
        o = V3(r0.x, r0.y, r0.z); d = V3(r1.x, r1.y, r1.z); tmax = r0.w;
        return true;
    };
    auto done = [&](uint32_t e, const HitState& best, bool hit) { a.hits[e] = make_float4(best.t, best.u, best.v, __int_as_float(hit ? best.tri : -1)); };
    if (a.persistent) {
        walk_pool<false, STATS>(a.nodes, a.tris, s_lds, (int)a.stack_entries, a.overflow, PoolStream{a.count, a.cursors, 0u, a.sub_capacity, 8u, g, 0u, a.pool, 0u, 0u}, a.refill_min, a.tri_min,
                                a.lanes_log2, load, done, nn, nt, wn, wt);
    } else {
        const uint32_t n = a.count[g * CRT_COUNTER_STRIDE];
        const uint32_t first = c * a.pool;
        if (first >= n) return;
        const uint32_t last = first + a.pool < n ? first + a.pool : n;
        walk_pool<false, STATS>(a.nodes, a.tris, s_lds, (int)a.stack_entries, a.overflow, PoolStatic{g * a.sub_capacity + first, g * a.sub_capacity + last}, a.refill_min, a.tri_min,
                                a.lanes_log2, load, done, nn, nt, wn, wt);
    }
    if (STATS) flush_visit_totals(a.visit_totals, nn, nt, wn, wt);
}

// The deferred NEE occlusion tests (path_trace.fs:968) of a frame, all segments in ONE launch: region r of the queue holds the shadow rays
// segment r (of those that defer) emitted, 8 sub-queues each.  Walked by walk_pool — full waves from the first step, a pool's last eight rays
// on eight lanes each — either as a persistent grid drawing from all n_regions x 8 sub-queues (a.persistent, PoolStream) or one single-wave
// workgroup per pool of a.pool rays (a.pool = 64 and refill_min = 65: one lock-step batch).  An OCCLUDED ray clears the visibility word of
// its contribution slot (queue entry: (o, tmax) (d, slot)); k_fold_paths then adds what is left, in segment order.  Same rays, same walks as
// the in-place form: occlusion and visit totals keep the oracle's values.
#ifndef CRT_SHADOW_OCC
#define CRT_SHADOW_OCC 8      // 64 VGPRs, no vector spill: d4 +0.8 .. 1.2 %, d2 +1.5 .. 2.1 % against the 76 registers of six waves (profiles/r05_experiments.md §8)
#endif
template <bool STATS>
__global__ void __launch_bounds__(64, CRT_SHADOW_OCC) k_shadow_deferred(ShadowArgs a) {
    extern __shared__ uint2 s_lds[];
    const uint32_t g = blockIdx.x & 7u, q = blockIdx.x >> 3;
    uint32_t nn = 0, nt = 0, wn = 0, wt = 0;
    uint32_t slot_of = 0;
    auto load = [&](uint32_t e, vec3& o, vec3& d, float& tmax) {        // e: flat entry, (region * 8 + sub-queue) * sub_capacity + place
        if (a.perm) e = a.perm[e];                                        // sorted: e was a place in the sorted order
        const float4 r0 = a.shadow[2 * (size_t)e], r1 = a.shadow[2 * (size_t)e + 1];
        o = V3(r0.x, r0.y, r0.z); d = V3(r1.x, r1.y, r1.z); tmax = r0.w;
        slot_of = __float_as_uint(r1.w);
        return true;
    };
    auto done = [&](uint32_t, const HitState&, bool occluded) { if (occluded) reinterpret_cast<float*>(a.contrib + slot_of)[3] = 0.0f; };
    if (a.perm) {
        // sorted by origin cell: the sorted array in eight equal parts, one per XCD group (k_nee_scan wrote their lengths), a persistent grid
        const uint32_t eighth = a.sort_meta[8u * CRT_COUNTER_STRIDE];
        walk_pool<true, STATS>(a.nodes, a.tris, s_lds, (int)a.stack_entries, a.overflow, PoolStream{a.sort_meta, a.cursors, 0u, eighth, 8u, g, 0u, a.pool, 0u, 0u},
                               a.refill_min, a.tri_min, a.lanes_log2, load, done, nn, nt, wn, wt);
    } else if (a.persistent) {
        // the wave starts on its own XCD group's sub-queue of region (q mod n_regions)
        const uint32_t q0 = (q % a.n_regions) * 8u + g;
        walk_pool<true, STATS>(a.nodes, a.tris, s_lds, (int)a.stack_entries, a.overflow, PoolStream{a.count, a.cursors, a.count_stride, a.sub_capacity, a.n_regions * 8u, q0, 0u, a.pool, 0u, 0u},
                               a.refill_min, a.tri_min, a.lanes_log2, load, done, nn, nt, wn, wt);
    } else {
        const uint32_t r = q / a.pools_per_region, c = q - r * a.pools_per_region;     // region (segment), pool within the group's sub-queue
        const uint32_t n = a.count[(size_t)r * a.count_stride + g * CRT_COUNTER_STRIDE];
        const uint32_t first = c * a.pool;
        if (first >= n) return;
        const uint32_t last = first + a.pool < n ? first + a.pool : n;
        const uint32_t qb = (r * 8u + g) * a.sub_capacity;
        walk_pool<true, STATS>(a.nodes, a.tris, s_lds, (int)a.stack_entries, a.overflow, PoolStatic{qb + first, qb + last}, a.refill_min, a.tri_min, a.lanes_log2, load, done,
                               nn, nt, wn, wt);
    }
    if (STATS) flush_visit_totals(a.visit_totals, nn, nt, wn, wt);
}

// sum[p] = (((sum[p] + L_0[p]) + L_1[p]) + ...): what n consecutive frames would have added, in their order, zero radiance skipped as
// add_to_sum's callers skip it.  L_k = l_final[k][p] as the path left it, or — a path that ended in a segment whose shadow rays are deferred
// (bit 31 of the w word) — that prefix plus the path's contribution slots in segment order: L = L + C_b for every slot b of its mask whose
// visibility word k_shadow_deferred left standing (the in-place form's `if (!occluded) L = L + C`, path_trace.fs:998, and its emitter adds).
__global__ void __launch_bounds__(256) k_fold_paths(float* __restrict__ sum, const float4* __restrict__ l_final, const float4* __restrict__ contrib, uint32_t n_pixels,
                                                    uint32_t n_samples, uint32_t first_slot_segment) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    const size_t n_paths = (size_t)n_samples * n_pixels;
    for (uint32_t smp = 0; smp < n_samples; ++smp) {
        const size_t path = (size_t)smp * n_pixels + p;
        const float4 l = l_final[path];
        vec3 L = V3(l.x, l.y, l.z);
        uint32_t m = __float_as_uint(l.w);
        if (m & 0x80000000u) {
            m = (m & 0xffffu) >> first_slot_segment;             // bit k = the slot of segment first_slot_segment + k
            for (uint32_t k = 0; m != 0u; ++k, m >>= 1)
                if (m & 1u) {
                    const float4 c = contrib[(size_t)k * n_paths + path];
                    if (c.w != 0.0f) L = L + V3(c.x, c.y, c.z);
                }
        }
        if (L.x != 0.f || L.y != 0.f || L.z != 0.f) add_to_sum(sum, p, L);
    }
}

// packed tile-major -> linear frame (bottom row first); pixels of other ranks stay untouched.
__global__ void __launch_bounds__(256) k_untile(FrameArgs f, const float* __restrict__ packed, float* __restrict__ linear) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < f.n_local_pixels; i += gridDim.x * blockDim.x) {
        uint32_t px, py;
        if (!pixel_of(f, i, px, py)) continue;
        const size_t o = 3 * ((size_t)py * f.width + px);
        linear[o] = packed[3 * (size_t)i];
        linear[o + 1] = packed[3 * (size_t)i + 1];
        linear[o + 2] = packed[3 * (size_t)i + 2];
    }
}

// Shader/output.fs:9-20: c = S / frames; c *= 1 / (1 + lum(c) / 2); pow(c, 1 / 2.2); 8-bit UNORM write.  The power is PINNED: pow's last
// bits differ between libm and the device, and the byte only depends on which of 255 thresholds the argument has reached — thr[j] = the
// smallest float x whose reference byte (double-precision pow, rounded once) is >= j, a table the host computes (crt_device.cpp
// gamma_thresholds; the oracle has its own) — so the byte is a count of thresholds: eight compares, and exact on both sides.
__device__ __forceinline__ uchar4 resolve_pixel(float s0, float s1, float s2, float inv_count, const float* __restrict__ thr) {
    const float c0 = s0 * inv_count, c1 = s1 * inv_count, c2 = s2 * inv_count;
    const float lum = 0.3f * c0 + 0.6f * c1 + 0.1f * c2;
    const float k = rcp_ieee(1.0f + __fdiv_rn(lum, 2.0f));
    const float cc[3] = {c0, c1, c2};
    uchar4 out;
    uint8_t* o8 = reinterpret_cast<uint8_t*>(&out);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float x = cc[ch] * 1.0f * k;
        uint32_t lo = 0u;                                  // thr[1..255] ascending; byte = the largest j with x >= thr[j] (0: none; NaN and negatives: none)
#pragma unroll
        for (uint32_t step = 128u; step != 0u; step >>= 1)
            if (x >= thr[lo + step]) lo += step;
        o8[ch] = (uint8_t)lo;
    }
    o8[3] = 255;
    return out;
}
__global__ void __launch_bounds__(256) k_resolve(const float* __restrict__ linear, uint32_t n_pixels, float inv_count, const float* __restrict__ thr,
                                                 uint8_t* __restrict__ rgba) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += gridDim.x * blockDim.x)
        reinterpret_cast<uchar4*>(rgba)[i] = resolve_pixel(linear[3 * (size_t)i], linear[3 * (size_t)i + 1], linear[3 * (size_t)i + 2], inv_count, thr);
}
// The same straight from a packed tile-major sum buffer (this device's own, or a peer's slice as gathered): un-tile and resolve in one pass —
// what the drop-in frame loop runs per displayed frame (Scene.h:1226-1230), 12 B read + 4 B written per pixel.
__global__ void __launch_bounds__(256) k_resolve_packed(FrameArgs f, const float* __restrict__ packed, float inv_count, const float* __restrict__ thr,
                                                        uint8_t* __restrict__ rgba) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < f.n_local_pixels; i += gridDim.x * blockDim.x) {
        uint32_t px, py;
        if (!pixel_of(f, i, px, py)) continue;
        reinterpret_cast<uchar4*>(rgba)[(size_t)py * f.width + px] = resolve_pixel(packed[3 * (size_t)i], packed[3 * (size_t)i + 1], packed[3 * (size_t)i + 2], inv_count, thr);
    }
}

// ------------------------------------------------------------------ launchers --------

// Timing events ride on the dispatch itself (hipExtLaunchKernelGGL): the events take the kernel's own start/stop
// timestamps, with none of the marker packets a hipEventRecord pair puts between back-to-back kernels (~4 us per pair).
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
void set_launch_events(hipEvent_t start, hipEvent_t stop) { g_ev_start = start; g_ev_stop = stop; }
template <typename K, typename A>
static inline void launch(K kernel, dim3 g, dim3 b, size_t lds, hipStream_t stream, const A& a) {
    // measurement aid: CRT_LDS_PAD=<bytes> adds unused dynamic LDS to every launch (LDS is handed out in 1,280-byte units: a way to take waves
    // off a CU without touching the kernel)
    static const size_t pad = [] { const char* e = getenv("CRT_LDS_PAD"); return e ? (size_t)strtoul(e, nullptr, 10) : (size_t)0; }();
    lds += pad;
    if (g_ev_start || g_ev_stop) {
        hipExtLaunchKernelGGL(kernel, g, b, (uint32_t)lds, stream, g_ev_start, g_ev_stop, 0, a);
        g_ev_start = g_ev_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kernel, g, b, lds, stream, a);
    }
}

// Workgroup shape of the traversal kernels: `waves` = 1 (every wave its own workgroup: the hardware dispatcher refills
// SIMDs wave by wave; measured 1 M mesh k_segment 0.246 -> 0.226 ms, Cornell 0.079 -> 0.074 ms) or 4 (256 threads).
// It is a per-scene setting handed to every launcher.  A request whose LDS stacks would not fit the 64 KB a kernel may
// ask for without raising its dynamic-LDS limit (only the BVH2 stack of a very deep tree at 4 waves gets there) falls
// back to single-wave workgroups, which the kernels' index mapping (wave_id) supports for any grid.
static inline uint32_t fit_waves(uint32_t waves, size_t lds_per_wave) {
    waves = waves == 1u ? 1u : waves == 2u ? 2u : 4u;
    return (size_t)waves * lds_per_wave > 64u * 1024u ? 1u : waves;
}
static inline size_t stack_bytes(uint32_t entries) { return (size_t)(entries + CRT_HIT_SLOTS) * 64 * sizeof(uint2); }   // the lane's stack column + its hit record
// `grid` counts 4-wave chunks; with single-wave workgroups each of them becomes 4 workgroups
static inline dim3 grid_dim(uint32_t grid, uint32_t waves) { return dim3(grid * (4u / waves)); }
static inline dim3 block_dim(uint32_t waves) { return dim3(waves * 64u); }

void launch_trace(const TraceArgs& a_in, int mode, bool stats, uint32_t grid, uint32_t waves, hipStream_t stream) {
    waves = fit_waves(waves, stack_bytes(a_in.stack_entries));
    TraceArgs a = a_in;
    if (waves != 1u || a.pool_split_log2 > 2u) a.pool_split_log2 = 0u;       // sub-pools are single-wave workgroups
    const dim3 g = grid_dim(grid << a.pool_split_log2, waves), b = block_dim(waves);
    const size_t lds = waves * stack_bytes(a.stack_entries);
    if (mode == 1) {
        if (stats) launch(k_trace<true, true>, g, b, lds, stream, a);
        else       launch(k_trace<true, false>, g, b, lds, stream, a);
    } else {
        if (stats) launch(k_trace<false, true>, g, b, lds, stream, a);
        else       launch(k_trace<false, false>, g, b, lds, stream, a);
    }
}
void launch_trace_bvh2(const Bvh2Args& a, int any, bool stats, uint32_t grid, uint32_t waves, hipStream_t stream) {
    const size_t per_wave = (size_t)a.stack_entries * 64 * sizeof(int);
    waves = fit_waves(waves, per_wave);
    const dim3 g = grid_dim(grid, waves), b = block_dim(waves);
    const size_t lds = waves * per_wave;
    if (any) {
        if (stats) launch(k_trace_bvh2<true, true>, g, b, lds, stream, a);
        else       launch(k_trace_bvh2<true, false>, g, b, lds, stream, a);
    } else {
        if (stats) launch(k_trace_bvh2<false, true>, g, b, lds, stream, a);
        else       launch(k_trace_bvh2<false, false>, g, b, lds, stream, a);
    }
}
// The first-segment kernels exist twice: compiled for 5 waves per SIMD (96 VGPRs) and for 6 (80 VGPRs).  A launch that fills the chip several
// times over runs faster with the sixth wave (CRT_SEG_OCC_FIRST); one that is only as long as its longest waves — a shard, a small frame,
// the side-by-side sample form — runs those waves faster at 5 (1/8 of a 1080p frame, 4 samples side by side: 0.0453 ms per frame at 5 waves,
// 0.0503 at 6).  The host says which (SegmentArgs::wide_first); counting kernels keep the one form.
// Feature level of the shading code: PLAIN (Lambert, untextured: the reference's shipped scene), MAT (+ mirror / Disney), FULL (+ textured
// albedo).  Counting kernels, the BVH2 frame mode, the deferred-shadow form and the shade-only (PRETRACED) form exist at the highest level
// only (the extra branches are decided per material at run time: same arithmetic, same sums) — 33 instantiations of k_segment.
//               k_segment<FIRST, STATS, TEX, PRETRACED, INPLACE, BVH2, MAT, BATCH, WIDE, ONE>
#define CRT_K(F, S, T, P, Y, B2, M, BA, WI, ON) k_segment<F, S, T, P, Y, B2, M, BA, WI, ON>
// returns bit 0: the launch ran the 6-waves-per-SIMD (WIDE) build of the first-segment kernel; bit 1: a one-pass (ONE) build (crt_debug_launch_info)
int launch_segment(const SegmentArgs& a, bool first, bool pretraced, bool inplace, bool bvh2, bool mat, bool stats, uint32_t grid, uint32_t waves, hipStream_t stream) {
    const bool tex = a.textures != nullptr;
    // one LDS region serves the CWBVH stack (8 B per level and lane) or the BVH2 stack (4 B)
    const size_t per_wave = std::max(stack_bytes(a.stack_entries), bvh2 ? (size_t)a.stack_entries2 * 64 * sizeof(int) : (size_t)0);
    waves = fit_waves(waves, per_wave);
    const dim3 g = grid_dim(grid, waves), b = block_dim(waves);
    const size_t lds = waves * per_wave;
    const int feat = tex ? 2 : mat ? 1 : 0;
    if (first && a.n_samples > 1u) {
        // several samples per launch (crt_render_frames): CWBVH, shadow rays in place, no counting (crt_device.cpp batch_limit)
        // wave_samples: as few passes as 4 waves allow, and no more waves than those passes need (5 samples: 2 passes of 3 waves)
        const uint32_t ws_passes = (a.n_samples + 3u) / 4u, ws = (a.n_samples + ws_passes - 1u) / ws_passes;
        const size_t lds4 = ws * per_wave + ws * 64 * sizeof(float4);
        SegmentArgs v = a;
        if (a.wave_samples && lds4 > 64u * 1024u) v.wave_samples = 0u;     // the sequential form, should the stacks and the result strip not fit
        if (v.wave_samples == 2u && ((a.n_samples & 3u) || waves != 1u)) v.wave_samples = 0u;   // samples in lanes come four at a time, on single-wave workgroups
        const bool side_by_side = v.wave_samples == 1u, in_lanes = v.wave_samples == 2u;
        // the samples on the 2 to 4 waves of a workgroup, one batch per workgroup: `grid` chunks of 4 batches = 4 * grid workgroups;
        // in the lanes of four single-wave workgroups per batch (one 4 x 4 pixel quadrant x 4 samples each): 16 * grid workgroups
        const dim3 gg = side_by_side ? dim3(grid * 4u) : in_lanes ? dim3(grid * 16u) : g, bb = side_by_side ? dim3(ws * 64u) : in_lanes ? dim3(64u) : b;
        const size_t ll = side_by_side ? lds4 : in_lanes ? per_wave : lds;
        const bool one_pass = in_lanes && a.n_samples == 4u && a.tri_min != 0u && a.lanes_log2 != 0u;      // four samples in the lanes of a wave: the builds without a sample loop
        // counting in the TIMED form (option count_visits 2): what the uniform node steps see depends on which rays share a wave
        if (stats && one_pass) { launch(CRT_K(true, true, true, false, true, false, true, true, false, true), gg, bb, ll, stream, v); return 2; }
        if (feat == 2 && one_pass)      { launch(CRT_K(true, false, true, false, true, false, true, true, false, true), gg, bb, ll, stream, v); return 2; }
        else if (feat == 1 && one_pass) { launch(CRT_K(true, false, false, false, true, false, true, true, false, true), gg, bb, ll, stream, v); return 2; }
        else if (feat == 2) launch(CRT_K(true, false, true, false, true, false, true, true, false, false), gg, bb, ll, stream, v);
        else if (feat == 1) launch(CRT_K(true, false, false, false, true, false, true, true, false, false), gg, bb, ll, stream, v);
        else if (v.wide_first && one_pass) { launch(CRT_K(true, false, false, false, true, false, false, true, true, true), gg, bb, ll, stream, v); return 3; }
        else if (v.wide_first && !side_by_side) { launch(CRT_K(true, false, false, false, true, false, false, true, true, false), gg, bb, ll, stream, v); return 1; }
        else                launch(CRT_K(true, false, false, false, true, false, false, true, false, false), gg, bb, ll, stream, v);
        return 0;
    }
    if (bvh2) {                                          // the shipped shader's own walks as a frame renderer: Lambert (+ textures) only
        if (first) { if (stats) launch(CRT_K(true, true, true, false, true, true, false, false, false, false), g, b, lds, stream, a);
                     else       launch(CRT_K(true, false, true, false, true, true, false, false, false, false), g, b, lds, stream, a); }
        else       { if (stats) launch(CRT_K(false, true, true, false, true, true, false, false, false, false), g, b, lds, stream, a);
                     else       launch(CRT_K(false, false, true, false, true, true, false, false, false, false), g, b, lds, stream, a); }
        return 0;
    }
    if (pretraced) {                                     // shade-only pass behind k_closest_queue (option "bounce_refill")
        if (inplace) { if (stats) launch(CRT_K(false, true, true, true, true, false, true, false, false, false), g, b, lds, stream, a);
                       else       launch(CRT_K(false, false, true, true, true, false, true, false, false, false), g, b, lds, stream, a); }
        else         { if (stats) launch(CRT_K(false, true, true, true, false, false, true, false, false, false), g, b, lds, stream, a);
                       else       launch(CRT_K(false, false, true, true, false, false, true, false, false, false), g, b, lds, stream, a); }
        return 0;
    }
    if (!inplace) {                                      // shadow rays deferred to k_shadow_deferred (option "inplace_shadow" 0 / 2)
        if (first) { if (stats) launch(CRT_K(true, true, true, false, false, false, true, false, false, false), g, b, lds, stream, a);
                     else       launch(CRT_K(true, false, true, false, false, false, true, false, false, false), g, b, lds, stream, a); }
        else if (stats) launch(CRT_K(false, true, true, false, false, false, true, false, false, false), g, b, lds, stream, a);
        else if (feat == 2) launch(CRT_K(false, false, true, false, false, false, true, false, false, false), g, b, lds, stream, a);
        else if (feat == 1) launch(CRT_K(false, false, false, false, false, false, true, false, false, false), g, b, lds, stream, a);
        else                launch(CRT_K(false, false, false, false, false, false, false, false, false, false), g, b, lds, stream, a);
        return 0;
    }
    if (stats) {
        if (first) launch(CRT_K(true, true, true, false, true, false, true, false, false, false), g, b, lds, stream, a);
        else       launch(CRT_K(false, true, true, false, true, false, true, false, false, false), g, b, lds, stream, a);
        return 0;
    }
    if (feat == 2)      { if (first) launch(CRT_K(true, false, true, false, true, false, true, false, false, false), g, b, lds, stream, a);
                          else       launch(CRT_K(false, false, true, false, true, false, true, false, false, false), g, b, lds, stream, a); }
    else if (feat == 1) { if (first) launch(CRT_K(true, false, false, false, true, false, true, false, false, false), g, b, lds, stream, a);
                          else       launch(CRT_K(false, false, false, false, true, false, true, false, false, false), g, b, lds, stream, a); }
    else                { if (first) launch(CRT_K(true, false, false, false, true, false, false, false, false, false), g, b, lds, stream, a);
                          else       launch(CRT_K(false, false, false, false, true, false, false, false, false, false), g, b, lds, stream, a); }
    return 0;
}
#undef CRT_K
// grid: 8 x (pools a sub-queue can hold) single-wave workgroups, a pool beyond its sub-queue's count returns at once; persistent: `grid_waves`
// workgroups (a multiple of 8), as many as the chip holds
void launch_closest_queue(const QueueTraceArgs& a, bool stats, uint32_t grid_waves, hipStream_t stream) {
    const dim3 g(grid_waves), b(64u);
    if (stats) launch(k_closest_queue<true>, g, b, stack_bytes(a.stack_entries), stream, a);
    else       launch(k_closest_queue<false>, g, b, stack_bytes(a.stack_entries), stream, a);
}
void launch_nee_sort(const NeeSortArgs& a, uint32_t blocks, hipStream_t stream) {
    hipLaunchKernelGGL(k_nee_hist, dim3(blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_nee_scan, dim3(1), dim3(1024), 0, stream, a);
    hipLaunchKernelGGL(k_nee_scatter, dim3(blocks), dim3(256), 0, stream, a);
}
void launch_shadow_deferred(const ShadowArgs& a, bool stats, uint32_t grid_waves, hipStream_t stream) {
    const dim3 g(grid_waves), b(64u);
    if (stats) launch(k_shadow_deferred<true>, g, b, stack_bytes(a.stack_entries), stream, a);
    else       launch(k_shadow_deferred<false>, g, b, stack_bytes(a.stack_entries), stream, a);
}
void launch_bin_scan(const BinScanArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, stream, a);
}
void launch_fold_paths(float* sum, const float4* l_final, const float4* contrib, uint32_t n_pixels, uint32_t n_samples, uint32_t first_slot_segment, hipStream_t stream) {
    hipLaunchKernelGGL(k_fold_paths, dim3((n_pixels + 255u) / 256u), dim3(256), 0, stream, sum, l_final, contrib, n_pixels, n_samples, first_slot_segment);
}
void launch_untile(const FrameArgs& f, const float* packed, float* linear, uint32_t grid, hipStream_t stream) {
    hipLaunchKernelGGL(k_untile, dim3(grid), dim3(256), 0, stream, f, packed, linear);
}
void launch_resolve(const float* linear, uint32_t n_pixels, float inv_count, const float* thr, uint8_t* rgba, uint32_t grid, hipStream_t stream) {
    hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(256), 0, stream, linear, n_pixels, inv_count, thr, rgba);
}
void launch_resolve_packed(const FrameArgs& f, const float* packed, float inv_count, const float* thr, uint8_t* rgba, uint32_t grid, hipStream_t stream) {
    hipLaunchKernelGGL(k_resolve_packed, dim3(grid), dim3(256), 0, stream, f, packed, inv_count, thr, rgba);
}

// crt_warmup: load this translation unit's code object on the current device (asking for a kernel's attributes does that without
// launching anything); the other kernels of the unit come with it
int warm_rt_kernels() {
    hipFuncAttributes a;
    hipError_t e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_trace<false, false>))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_trace<true, false>))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_shadow_deferred<false>))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_untile))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_resolve))) != hipSuccess) return (int)e;
    if ((e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_fold_paths))) != hipSuccess) return (int)e;
    return 0;
}

}  // namespace crt
