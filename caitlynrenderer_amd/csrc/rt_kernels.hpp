// Kernel argument blocks and launcher prototypes shared by rt_kernels.hip and crt_device.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crt {

// Upper bound of traversal-stack entries per ray, kept in LDS (reference LOCAL_STACK_SIZE, cwbvh.fs:374).
// crt_scene_create refuses a CWBVH deeper than this; the launch uses min(this, depth of the tree).
#define CRT_STACK_ENTRIES 16
// Two more entries behind every lane's stack column: (u, v) and the original id of a closest-hit walk's best hit so far — written a few
// times per ray, read once: LDS instead of three VGPRs across the traversal loop (rt_kernels.hip walk_pool, walk_batch; the id's
// neighbour word is walk_batch's regroup scratch).
#ifndef CRT_HIT_SLOTS
#define CRT_HIT_SLOTS 2
#endif
// Rows of 16 bytes between two nodes / two triangle records of the DEVICE copies the traversal reads (the C ABI's crt_node8 is 5 rows,
// a record 3).  5 / 3 = packed.  8 / 4 pads a node to 128 bytes and a record to 64, so that neither straddles a cache line: a scene
// that misses the caches then fetches one line per node instead of 1.6 on average (profiles/r03_experiments.md, "line-aligned records").
#ifndef CRT_NODE_ROWS
#define CRT_NODE_ROWS 5
#endif
#ifndef CRT_TRI_ROWS
#define CRT_TRI_ROWS 3
#endif
#define CRT_TRACE_BLOCK 256

struct TraceArgs {
    const uint4* nodes;        // 5 x 16 B per node8 (cwbvh.fs:484-488)
    const float4* tris;        // 3 x 16 B per CWBVH triangle: (v0|orig_id) (e1|bvh2 slot) (e2|material)
    const float4* rays;        // crt_ray: (o, tmax) (d, payload)
    float4* hits;              // crt_hit: (t, u, v, tri)
    uint32_t* stats;           // optional: nodes | tris << 16
    const uint32_t* count_ptr; // device-side ray count, or null
    uint32_t n;
    uint32_t out_orig_id;      // 1: hit.tri = original triangle id, 0: CWBVH triangle index
    uint32_t stack_entries;
    uint32_t refill_min;       // idle lanes that trigger a pool refill (walk_pool)
    uint32_t tri_min;          // vote ratio of the walks: node step while node-ready lanes >= tri_min x triangle-waiting lanes
    uint32_t* overflow;        // += 1 per dropped stack push (never happens for a tree crt_scene_create accepted)
    uint32_t pool_split_log2;  // a wave's 256-ray slot of the index space is walked by 1 << this single-wave workgroups (pools of 256, 128, 64)
};

struct Bvh2Args {               // the reference's live BVH2 walk (path_trace.fs:511-819) for crt_trace
    const float4* nodes;       // FlatNode: (bmin.xyz, link) (bmax.xyz, count), Scene.h:1057-1062
    const float4* tris;        // 3 x 16 B per BVH2 leaf slot: (v0|orig id) (e1|slot) (e2|material)
    const float4* rays;
    float4* hits;
    uint32_t* stats;
    uint32_t n;
    uint32_t tie;              // 0: first visited wins (path_trace.fs:363), 1: lowest original id
    uint32_t stack_entries;    // >= BVH2 depth + 1
    uint32_t* overflow;
};

struct PathBuffers {           // indexed by path = local pixel (+ sample * n_local_pixels in a batched launch); touched only by paths longer than one segment
    float4* L;                 // radiance so far, prev_pdf
    float4* T;                 // throughput, is_specular
    float2* seed;              // shader RNG state (path_trace.fs:27)
};

struct FrameArgs {
    const uint2* tile_xy;      // local tile -> (tile x, tile y)
    const uint32_t* tile_order; // processing slot -> local tile (which tile's pixels the k-th unit of work renders); null = identity.
                               // Storage (sum buffer, path state, the gather between ranks) is always by local tile.
    uint32_t n_local_pixels;   // n_local_tiles * tile * tile
    uint32_t tile, width, height;
    uint32_t tile_log2;        // log2(tile) when the tile is a power of two (shifts instead of integer divisions), else 0
    uint32_t jitter;
    float rv;                  // randomVector.x * randomVector.y (path_trace.fs:40)
    float tan_fov, aspect_tan; // tan(fov/2), W/H*tan(fov/2) (path_trace.fs:1041-1043)
    float cam_pos[3], cam_right[3], cam_up[3], cam_forward[3];
};

// Bounce rays regrouped between two segments (BASELINE configs[3], "divergence / sorting stress"): the rays a launch emits are
// appended to one of CRT_RAY_BINS bins keyed by (direction octant, 8 x 8 x 8 cell of the origin) instead of to its group's
// sub-queue, so that the next launch's 64-ray batches hold rays that start in the same region and head the same way.  No sort
// pass: every bin has a fixed place in the queue whose size is what the bin received in the PREVIOUS launch of the same segment
// (+ 1/8 + 16; a progressive renderer repeats its distribution frame after frame), a ray that does not fit goes to an overflow
// region behind the bins (all of them in the first launch of a scene), and one tiny kernel between the two launches turns the
// bins' fill counts into the consumer's index space and next frame's capacities (k_bin_scan).  Per-ray work is unchanged and a
// pixel's radiance is added in frame order as before, so sums, ray counts and visit counters keep their bits.
#define CRT_RAY_BINS 4096u
struct RayBins {
    uint32_t* count;            // [CRT_RAY_BINS] rays this launch appended per bin (also those that overflowed); null = sub-queues
    const uint32_t* cap;        // [CRT_RAY_BINS] places of each bin in the queue
    const uint32_t* off;        // [CRT_RAY_BINS] first queue entry of each bin
    uint32_t* ovf_count;        // rays that found their bin full: entry ovf_base + k
    uint32_t ovf_base;
    uint32_t per_lane;          // how the rays of a wave that share a key find each other.  0: one ballot per distinct key, one atomic per key
                                // (coherent first segment: a handful of keys); 1: not at all, every ray takes its place with its own
                                // atomic; 2: by ranking through a wave-private LDS table hashed by the key, one atomic per key (waves of
                                // bounce rays: ~50 keys)
    float origin[3], scale[3];  // cell coordinate = clamp((p - origin) * scale, 0, 7)
};
struct BinScanArgs {             // k_bin_scan: one workgroup of 1024 threads between the emitting and the consuming launch
    uint32_t* count;             // in: fill attempts per bin; zeroed for the next launch
    const uint32_t* cap;         // in: the capacities the emitting launch used
    uint32_t* start;             // out [CRT_RAY_BINS + 1]: first consumer index of each bin; [last] = rays that sit in bins
    uint32_t* ovf_count;         // in: overflow entries; zeroed
    uint32_t* n_in;              // out: rays the consuming launch will find (in bins + overflow)
    uint32_t* cap_next;          // out: capacities and offsets for the next launch that emits into this segment
    uint32_t* off_next;
    uint32_t queue_entries;      // places available to the bins (the overflow region lies behind them)
};

struct SegmentArgs {
    const uint4* nodes;
    const float4* planes;      // 12 rows per node: its child planes as floats (uniform node steps)
    const float4* tris;
    const int4* triangles;     // 3 x int4 per BVH2-ordered triangle (v, vn, vt)
    const float* normals;
    const float4* materials;
    const float* lights;
    int32_t n_lights;
    uint32_t stack_entries;
    const float2* texcoords;   // uv pairs (Scene.h:1029-1034)
    const float* textures;     // albedo array as RGB32F = c/255 (Scene.h:1065-1078), layer-major; null when absent
    int32_t tex_width, tex_height, n_textures;
    FrameArgs f;
    uint32_t sub_capacity;     // entries per sub-queue (8 sub-queues per queue)
    uint32_t tri_min;          // vote ratio of the walks; 0 = plain per-lane loop (tiny trees)
    uint32_t lanes_log2;       // walk_batch: a ray may spread over up to 1 << lanes_log2 lanes as its wave drains (0: one lane per ray throughout)
    const float4* rays_in;     // segments >= 1: crt_ray with payload = local pixel
    const uint32_t* count_in;  // 8 counters, CRT_COUNTER_STRIDE apart
    const float4* hits_in;     // k_segment<PRETRACED>: (t, u, v, CWBVH triangle) per queue entry
    float4* rays_next;   uint32_t* count_next;
    float4* shadow;      uint32_t* count_shadow;   // this segment's shadow rays: their count (8 counters) and, when they are DEFERRED (k_segment<!INPLACE>), this
                                                   // segment's region of the NEE queue, 2 x float4 per entry: (o, tmax) (d, contribution slot)
    float4* contrib;           // deferred: this segment's contribution slots, one per path: (C | T e, visibility word 1.0 / 0.0)
    uint32_t slot_first;       // deferred: index of this segment's slot of path 0 in the frame's slot array (a queue entry names slot_first + path)
    uint32_t slot_bit;         // deferred: this segment's bit in the path state's slot mask (1 << (8 + segment))
    PathBuffers pb;
    float* sum;                // packed tile-major RGB32F
    uint32_t last_segment;
    unsigned long long* visit_totals;   // STATS: [0] += nodes, [1] += tris
    const float4* nodes2;      // BVH2 mode: FlatNode array and slot-ordered records (see Bvh2Args), int stack entries, tie rule
    const float4* tris2;
    uint32_t stack_entries2;
    uint32_t tie;
    uint32_t* zero_counts;     // FIRST: the other frame's counter bank, cleared here for the next frame (no memset launch)
    uint32_t n_zero;
    uint32_t* overflow;        // += 1 per dropped stack push
    float4* l_final;           // crt_render_frames on paths of several segments: where a finished path leaves its radiance, indexed like
                               // the path state by sample * n_local_pixels + pixel; k_fold_paths adds them to `sum` in sample order
    uint32_t* tile_cost;       // FIRST, optional: += the clock ticks every wave spent on a tile's pixels, per local tile (feeds tile_order)
    uint32_t wide_first;       // FIRST: 1 = the 6-waves-per-SIMD build of the kernel (launches bound by throughput), 0 = the 5-wave one
    uint32_t wave_samples;     // BATCH: 2 = a wave renders a 4 x 4 pixel quadrant of a batch x 4 samples (lane = sample * 16 + pixel; n_samples a
                               // multiple of 4, four times the waves), the samples of a pixel added in order through lane shuffles;
                               // 1 = the launch's samples (2..4) sit on the waves of a 4-wave workgroup, one sample of
                               // the workgroup's 64 pixels each, and are added to the sum in sample order through LDS; 0 = one wave
                               // renders its pixels' samples one after the other
    uint32_t n_samples;        // FIRST: samples per pixel rendered by this launch (>= 1); > 1 only for one-segment paths walked in place
    float rv_s[8];             // randomVector.x * randomVector.y of each of them (f.rv = rv_s[0])
    RayBins bins_out;          // where the rays this launch emits go (count == null: rays_next's sub-queues)
    const uint32_t* bin_start; // segments >= 1, rays_in binned: [CRT_RAY_BINS + 1] consumer index -> bin; null = sub-queues
    const uint32_t* bin_off_in;
    uint32_t ovf_base_in;
};

struct QueueTraceArgs {        // k_closest_queue: closest hit for a device-written path-ray queue
    const uint4* nodes;
    const float4* tris;
    const float4* rays;        // 8 sub-queues of crt_ray
    const uint32_t* count;
    float4* hits;              // parallel to the queue
    uint32_t stack_entries;
    uint32_t sub_capacity;
    uint32_t refill_min;
    uint32_t tri_min;
    uint32_t pool;             // rays per wave (walk_pool): 64 with refill_min 65 = one lock-step batch
    uint32_t lanes_log2;       // != 0: the pool's last eight rays get eight lanes each
    uint32_t persistent;       // 1: a grid of resident waves that draw rays from the sub-queues through `cursors` until all are dry
    uint32_t* cursors;         // persistent: one per sub-queue, CRT_COUNTER_STRIDE apart, zero at launch
    unsigned long long* visit_totals;
    uint32_t* overflow;
};

struct ShadowArgs {            // k_shadow_deferred: the deferred NEE shadow rays of a frame, all segments in one launch
    const uint4* nodes;
    const float4* tris;
    const float4* shadow;      // the NEE queue: [region][8 sub-queues][sub_capacity] x 2 float4 (o, tmax) (d, contribution slot)
    const uint32_t* count;     // region r, group g: count[r * count_stride + g * CRT_COUNTER_STRIDE]
    float4* contrib;           // the frame's contribution slots; an occluded ray clears its slot's visibility word
    uint32_t count_stride;
    uint32_t pools_per_region; // pools a sub-queue can hold: ceil(sub_capacity / pool)
    uint32_t stack_entries;
    uint32_t sub_capacity;
    uint32_t refill_min;
    uint32_t tri_min;
    uint32_t pool;
    uint32_t lanes_log2;
    uint32_t persistent;       // 1: a grid of resident waves drawing from all n_regions x 8 sub-queues through `cursors`
    uint32_t n_regions;
    uint32_t* cursors;         // persistent: one per (region, sub-queue), CRT_COUNTER_STRIDE apart, zero at launch
    const uint32_t* perm;      // sorted (option sort_shadow): place in the sorted order -> queue entry; null = emission order
    const uint32_t* sort_meta; // sorted: rays in each eighth of the sorted array [k * CRT_COUNTER_STRIDE], length of an eighth [8 * CRT_COUNTER_STRIDE]
    unsigned long long* visit_totals;
    uint32_t* overflow;
};

struct NeeSortArgs {           // k_nee_hist / k_nee_scan / k_nee_scatter: the frame's deferred shadow rays sorted by the cell they start in
    const float4* shadow;      // the NEE queue (ShadowArgs::shadow)
    const uint32_t* count;     // its counters, as ShadowArgs::count / count_stride
    uint32_t count_stride, sub_capacity, n_queues;      // n_queues = regions x 8
    float origin[3], scale[3]; // cell coordinate = clamp((p - origin) * scale, 0, 15)
    uint32_t* hist;            // [4096] zero between frames
    uint32_t* cursor;          // [4096] scratch
    uint32_t* perm;            // out: [total] queue entries in sorted order
    uint32_t* meta;            // out: see ShadowArgs::sort_meta
};

// waves = waves per workgroup: 1 (every wave its own workgroup), 2 or 4 (256 threads); a per-scene setting
void launch_trace(const TraceArgs& a, int mode, bool stats, uint32_t grid, uint32_t waves, hipStream_t stream);
void launch_trace_bvh2(const Bvh2Args& a, int any, bool stats, uint32_t grid, uint32_t waves, hipStream_t stream);
// mat: the scene has Mirror / Disney materials (CWBVH segments only: not with bvh2)
// inplace_shadow: the NEE shadow rays are walked inside the kernel; false = deferred to the frame's k_shadow_deferred launch
// returns bit 0: the launch ran the 6-waves-per-SIMD (WIDE) build of the first-segment kernel, bit 1: a one-pass (ONE) build
int launch_segment(const SegmentArgs& a, bool first, bool pretraced, bool inplace_shadow, bool bvh2, bool mat, bool stats, uint32_t grid, uint32_t waves, hipStream_t stream);
void launch_closest_queue(const QueueTraceArgs& a, bool stats, uint32_t grid_waves, hipStream_t stream);
void launch_shadow_deferred(const ShadowArgs& a, bool stats, uint32_t grid_waves, hipStream_t stream);
void launch_nee_sort(const NeeSortArgs& a, uint32_t blocks, hipStream_t stream);
// start/stop events for the NEXT traversal-kernel launch of this thread (either may be null); consumed by it
void set_launch_events(hipEvent_t start, hipEvent_t stop);
void launch_bin_scan(const BinScanArgs& a, hipStream_t stream);
// contrib may be null (no deferred segment); first_slot_segment = the segment slot 0 of the slot array belongs to
void launch_fold_paths(float* sum, const float4* l_final, const float4* contrib, uint32_t n_pixels, uint32_t n_samples, uint32_t first_slot_segment, hipStream_t stream);
void launch_untile(const FrameArgs& f, const float* packed, float* linear, uint32_t grid, hipStream_t stream);
// thr: the 256 thresholds of the pinned gamma (thr[j] = smallest x whose byte is >= j; thr[0] unused), in device memory
void launch_resolve(const float* linear, uint32_t n_pixels, float inv_count, const float* thr, uint8_t* rgba, uint32_t grid, hipStream_t stream);
void launch_resolve_packed(const FrameArgs& f, const float* packed, float inv_count, const float* thr, uint8_t* rgba, uint32_t grid, hipStream_t stream);

void set_step_hist(unsigned long long* d_hist, uint32_t mode = 0u);      // measurement aid: [2][65] node-step histogram of the counting kernels (mode 0: by enabled lanes, 1: by distinct nodes), null = off
int warm_rt_kernels();      // crt_warmup: loads this unit's code object on the current device

}  // namespace crt
