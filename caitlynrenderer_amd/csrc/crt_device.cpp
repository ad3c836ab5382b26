// Device half of include/crt.h: scene upload, per-frame wavefront dispatch, read-back.
// Replaces Scene::gpu_data / Render / update (Caitlyn/Scene.h:1000-1246).  Owns its own HIP
// stream; every kernel of a frame is enqueued there, queue lengths stay on the device, and the
// host synchronises once per call (never inside the frame).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <vector>

#include "../../include/crt.h"
#include "crt_error.hpp"
#include "device_build.hpp"
#include "host/cwbvh.hpp"
#include "host/flatnode_link.hpp"
#include "rt_kernels.hpp"

using crt::fail;

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(CRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

namespace {

constexpr int kMaxEvents = 8 + 8 * 16;
constexpr uint32_t kCounterStride = 32;                        // == CRT_COUNTER_STRIDE in rt_kernels.hip (128 B)
// kind 0 = rays into the segment, 1 = its shadow rays
constexpr uint32_t kQueueCounters = 2 * 17 * 8 * kCounterStride;    // (segment, kind, group) x stride
inline uint32_t counter_index(uint32_t seg, uint32_t kind, uint32_t group) { return ((seg * 2 + kind) * 8 + group) * kCounterStride; }
// behind them, in the same banks (cleared with them): the cursors of the persistent grids (rt_kernels.hip PoolStream) — one per sub-queue of
// a segment's path-ray queue (k_closest_queue), one per (region, sub-queue) of the frame's NEE queue (k_shadow_deferred)
inline uint32_t cursor_index_closest(uint32_t seg) { return kQueueCounters + seg * 8 * kCounterStride; }
constexpr uint32_t kCursorShadow = kQueueCounters + 17 * 8 * kCounterStride;
constexpr uint32_t kCounters = kCursorShadow + 17 * 8 * kCounterStride;

struct EventSpan { hipEvent_t a = nullptr, b = nullptr; int kind = 0; };   // kind: 0 raygen 1 closest 2 any 3 shade/other

template <typename T>
int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
    if (e != hipSuccess) return fail(CRT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return CRT_OK;
}

uint32_t morton2(uint32_t x, uint32_t y) {
    auto spread = [](uint32_t v) {
        v &= 0xffffu;
        v = (v | (v << 8)) & 0x00ff00ffu;
        v = (v | (v << 4)) & 0x0f0f0f0fu;
        v = (v | (v << 2)) & 0x33333333u;
        v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    return spread(x) | (spread(y) << 1);
}

}  // namespace

struct crt_scene {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    uint32_t width = 0, height = 0, max_depth = 1, n_lights = 0;

    // scene (replicated on every rank)
    uint4* d_nodes = nullptr;
    float4* d_planes = nullptr;          // the nodes' child planes as floats, 12 rows per node (uniform node steps; built by ensure_planes at the first CWBVH frame)
    float4* d_tris = nullptr;
    int4* d_triangles = nullptr;
    float* d_normals = nullptr;
    float4* d_materials = nullptr;
    float* d_lights = nullptr;
    float2* d_texcoords = nullptr;
    float* d_textures = nullptr;         // albedo array as RGB32F (c / 255.0f)
    int32_t tex_width = 0, tex_height = 0, n_textures = 0;
    crt_bvh_info info{};
    float4* d_bvh2 = nullptr;            // FlatNode array as uploaded by the reference (only when desc.bvh was given)
    float4* d_tris2 = nullptr;           // intersection records in BVH2 leaf-slot order
    uint32_t bvh2_stack = 0;             // BVH2 depth + 2

    // shard + frame buffers
    uint32_t rank = 0, world = 1, tile = 16;   // 16x16: four waves per tile — fine enough for the cost-sorted schedule (1 M triangles: 0.273 ms at 64, 0.257 at 16)
    std::vector<uint2> tiles;            // local tiles
    // Processing order of the local tiles (FrameArgs::tile_order): centre-out to begin with, then by measured cost, most
    // expensive first (longest-processing-time-first scheduling of the launch).  One frame after every change of camera,
    // shard or frame size has each wave add the clock ticks it spent to its tile's counter; the counters come back
    // through a pinned buffer without a stream synchronise, and the re-sorted order is uploaded in stream order.
    uint32_t* d_tile_order = nullptr; uint32_t* d_tile_cost = nullptr;
    uint32_t* h_tile_cost = nullptr; uint32_t* h_tile_order = nullptr;      // pinned
    hipEvent_t ev_tile_cost = nullptr, ev_tile_order = nullptr;
    enum { TILES_WANT = 0, TILES_PENDING = 1, TILES_DONE = 2 };
    int tile_state = TILES_WANT;
    uint32_t adaptive_tiles = 1;             // option: 0 keeps the centre-out order
    bool tile_order_uploading = false;
    bool tiles_measured_once = false;
    uint32_t frames_since_tile_measure = 0;
    bool capturing = false;                  // crt_debug_time_graph: no host-side decisions inside a stream capture
    uint2* d_tile_xy = nullptr;
    uint32_t n_local_tiles = 0, n_local_pixels = 0;
    uint64_t n_local_in_frame = 0;
    float* d_sum = nullptr;
    float* d_linear = nullptr;
    uint8_t* d_rgba = nullptr;
    uint8_t* h_rgba = nullptr;           // pinned staging buffer of crt_resolve
    float* d_gamma = nullptr;            // the 256 thresholds of the pinned gamma (resolve kernels)
    float4* d_rays[2] = {nullptr, nullptr};   // path-ray queues, only for max_depth > 1
    // DEFERRED NEE shadow rays (option "inplace_shadow" 0 / 2, rt_kernels.hip k_segment<!INPLACE>): the frame's NEE queue — one region per
    // deferring segment, 8 sub-queues each, 2 x float4 per ray: (o, tmax) (d, contribution slot) — and the contribution slots, one per
    // (deferring segment, path): (C | T e, visibility word)
    float4* d_nee = nullptr;
    float4* d_contrib = nullptr;
    // option "sort_shadow": the frame's deferred shadow rays walked in the order of the cell they start in (rt_kernels.hip k_nee_*):
    // perm[place] = queue entry, 4096-bin histogram + cursors + the eight parts' lengths
    uint32_t* d_nee_perm = nullptr;
    uint32_t* d_nee_bins = nullptr;           // hist [4096] | cursor [4096] | meta [9 x stride]
    uint32_t sort_shadow = 0;                 // (1 M triangles, four segments: 7.18 against 7.59 Gray/s unsorted — the sort costs more than the 4 % fewer any-hit node steps return)
    uint32_t defer_cap = 0, defer_regions = 0, defer_sub_capacity = 0;   // what the two were sized for
    uint32_t lfinal_cap = 0;                  // samples d_lfinal is sized for
    float4* d_qhits = nullptr;                // closest hits of the path-ray queue (max_depth > 1, option bounce_refill)
    // segments >= 1: 0 = fused lock-step k_segment (default); 1 = closest hits through lane-refill pools (k_closest_queue) + shade-only pass
    uint32_t bounce_refill = 0;
    uint32_t refill_pool = 256;               // rays per wave of k_closest_queue (64 with refill_min 65: lock-step batches)
    // k_shadow_deferred: rays per pool / per chunk a wave reserves, and the idle lanes that trigger a refill (65: never — one lock-step
    // batch per 64 rays of the pool).  1 M triangles, four segments: lock-step 64 7.34, static pools 128 / 16 7.42 (7.61 - 7.67 after the
    // kernarg change), persistent 256 / 16 7.72 Gray/s
    uint32_t shadow_pool = 256;
    uint32_t shadow_refill_min = 16;
    uint32_t step_hist_mode = 0;             // crt_debug_step_hist: 0 = node steps by enabled lanes, 1 = by distinct (node, octant) keys in the wave
    uint32_t shadow_waves = 6;               // waves per SIMD the persistent any-hit grid of the deferred shadow rays is sized for (the kernel fits 8; two shards'
                                             // grids share the chip: 4 / 5 / 6 / 7 / 8 = 7,723 / 7,767 / 7,776 / 7,750 / 7,721 on four segments, 10,540 / 10,450 / 10,338 at 5 / 6 / 8 on two)
    // the pool launches (k_shadow_deferred, k_closest_queue) as PERSISTENT grids: as many waves as the chip holds, each reserving chunks of
    // `pool` rays through per-queue cursors until every queue is dry; 0 = one workgroup per pool
    uint32_t persistent = 1;
    // crt_render_frames, how the samples of a launch sit on the hardware (include/crt.h, option "wave_samples"): 0 = one after the other in
    // each wave; 1 = side by side on the waves of a workgroup; 2 (default) = four samples of a 4 x 4 pixel quadrant in the lanes of a wave
    // where the launch allows it (a multiple of 4 samples, a tree of >= 64 nodes, CWBVH), otherwise 1 when the launch is bound by its longest
    // waves rather than by throughput (use_wave_samples) and 0 when not; 3 = lanes where allowed, else 0
    uint32_t wave_samples = 2;
    uint32_t wide_first = 2;            // first-segment kernels built for 6 waves per SIMD: 0 never, 1 always, 2 by the same measure
    float tile_cost_spread = 0.f;       // 99th percentile of the measured tile costs over their mean; 0 = nothing measured yet
    int last_launch_form = 0;           // crt_debug_launch_form
    int last_launch_wide = 0, last_launch_samples = 1, last_launch_one_pass = 0;      // crt_debug_launch_info
    bool use_wave_samples() const { return wave_samples == 2u ? bound_by_longest_waves() : wave_samples == 1u; }
    bool bound_by_longest_waves() const {
        // One wave renders the n samples of its 64 pixels one after the other: the launch cannot end before the most expensive
        // waves have done n samples, c99 * n, while the chip needs about mean * n * waves / slots for all of them.  When the first
        // is the larger, the samples go on 4 waves side by side (1 M triangles at 1080p, 8 frames per launch: 1/4 of the frame
        // 0.133 -> 0.062 ms per frame, 1/8 0.114 -> 0.039; the whole frame is throughput-bound and stays as it is, and so does a
        // Cornell box down to 1/4 of the frame — its waves are short and four times as many of them cost more than they save).
        const double waves = (double)(n_local_pixels + 63u) / 64.0, slots = (double)n_cu * 4.0 * 5.0;
        if (tile_cost_spread == 0.f) return waves <= 2.0 * slots;
        return (double)tile_cost_spread * slots >= 0.9 * waves;
    }
    crt::PathBuffers pb{};
    uint32_t stack_entries = CRT_STACK_ENTRIES;
    uint32_t sub_capacity = 0;                // entries per sub-queue (8 per queue)
    uint32_t* d_counts = nullptr;        // 2 banks (frame parity) of counter(seg, kind, group): kind 0 = rays into segment, 1 = its shadow rays
    uint32_t bank = 0;                   // bank of the most recent frame
    bool counts_clean = false;           // both banks known to be zero where the next frame needs them
    uint32_t* counts() const { return d_counts + (size_t)bank * kCounters; }
    uint32_t* h_counts = nullptr;        // pinned
    bool frame_buffers_ready = false;

    crt_camera cam{};
    bool have_camera = false;
    uint32_t jitter = 1;
    bool count_visits = false;
    bool count_batched = false;               // option count_visits 2: counting frames may share a launch in the timed form (four samples in the lanes of a wave)
    unsigned long long* d_visit_totals = nullptr;   // [0..3] lane visits: closest nodes/tris, any nodes/tris; [4..7] wave-level steps of the same blocks; [8] closest-hit rays that hit; [10], [11] closest / any-hit node visits of uniform node steps
    unsigned long long* h_visit_totals = nullptr;   // pinned

    // scratch for crt_trace (host rays)
    float4* d_t_rays = nullptr; float4* d_t_hits = nullptr; uint32_t* d_t_stats = nullptr; size_t t_cap = 0;

    // telemetry
    std::vector<EventSpan> spans;
    int n_spans = 0;
    crt_frame_stats stats{};
    bool stats_pending = false;
    bool stats_from_frame = false;
    bool stats_counted = false;
    uint32_t tri_min = 2;                    // traverse_pool vote: node step while node-ready lanes >= tri_min x triangle-waiting lanes
    // NEE shadow rays: 1 = walked inside k_segment, every segment; 0 = every segment's deferred to ONE any-hit launch behind the last
    // segment (k_shadow_deferred) with the contributions folded per path in segment order (k_fold_paths); 2 = first segment in place,
    // bounce segments deferred; 3 (default) = 2 for trees of 64+ nodes (1 M triangles, four segments: 7.23 -> 7.72 Gray/s), else 1
    uint32_t inplace_shadow = 3;
    uint32_t accel = 0;                      // frames: 0 CWBVH; 1 BVH2 walked as the shipped shader does (first visited wins); 2 BVH2, lowest id wins
    uint32_t refill_min = 8;                // walk_pool (crt_trace, k_closest_queue): idle lanes that trigger a refill
    // crt_trace: rays per wave (its refill pool): 64, 128 or 256.  64 = one lock-step batch per single-wave workgroup: four times the
    // workgroups for the dispatcher to balance, which is what a launch of a few million rays needs (tools/refill_probe.py, 1 M
    // triangles: 2.07 M primary rays 0.153 ms against 0.346 with 256-ray pools, whose 8,100 waves all start at once and end with the
    // slowest; 0.81 M shadow rays 0.192 / 0.202 / 0.230 ms for 64 / 128 / 256; 1.16 M bounce rays 0.332 / 0.320 / 0.321 — the one
    // case where refill beats the finer grain, by 4 %)
    uint32_t trace_pool = 64;
    uint32_t trace_occupancy = 8;            // persistent grids only (oversubscribe >= 1): workgroups per CU
    // 0: one chunk per workgroup, the hardware dispatcher hands chunks to CUs as they drain (measured 13 % faster than
    // a persistent grid on the 1 M mesh: per-chunk cost varies 10x between sky and grazing rays);
    // k >= 1: persistent grid of k x the resident workgroups, static schedule (rt_kernels.hip)
    uint32_t oversubscribe = 0;
    bool special_materials = false;          // some material is Mirror_type / Disney_type (albedo.w, Scene.h:111-132): k_segment<MAT>
    uint32_t waves_per_workgroup = 1;        // 1 = every wave its own workgroup (default), 2, or 4 = 256-thread workgroups
    uint32_t lanes_per_ray = 8;              // option "lanes_per_ray" (1, 2, 4, 8): how far a ray may spread over the lanes of its draining wave (rt_kernels.hip walk_batch)
    uint32_t* d_overflow = nullptr;          // dropped stack pushes since scene creation (stays 0 for every accepted tree)
    float4* d_lfinal = nullptr;              // batched frames on multi-segment paths: per (sample, pixel) final radiance (SegmentArgs::l_final)
    // bounce rays regrouped by (direction octant, origin cell) between segments (rt_kernels.hpp RayBins).  Tables per segment a ray
    // can enter (1..16), capacities / offsets twice (frame parity: the launch that consumes a segment's rays reads the layout its
    // producer used while k_bin_scan already writes the next frame's).
    // option "ray_bins": 0 (default) = per-group sub-queues in emission order; 1 = bins, the lanes of a wave that share a key take their
    // places with one atomic; 2 = bins, one atomic per ray; 3 = 1 for the first segment's emission, 2 for the bounce segments'.
    // Measured on the 1 M-triangle frame, 4 segments, 4 samples per launch: wave-level traversal steps -10 % (lane utilisation of
    // the closest-hit node block 46.5 -> 51.7 %), frame time +3.6 % / +31 % / +5 % for 1 / 2 / 3 — what the append costs
    // (a wave of bounce rays holds ~50 different keys) exceeds what the walk gains; profiles/r03_experiments.md.
    uint32_t ray_bins = 0;
    bool rows_padded = false;                // d_nodes / d_tris are at the build's stride (CRT_NODE_ROWS / CRT_TRI_ROWS)
    bool rays_doubled = false;               // the path-ray queues have their overflow half (allocated when the bins are first used)
    uint32_t* d_bins = nullptr;              // one allocation: count [17][B] | cap [2][17][B] | off [2][17][B] | start [17][B + 1] | ovf [17 x 32]
    float bounds_lo[3] = {0.f, 0.f, 0.f}, bounds_hi[3] = {1.f, 1.f, 1.f};   // of the vertices: the cell grid of the bins
    uint32_t* bin_count(uint32_t seg) const { return d_bins + (size_t)seg * CRT_RAY_BINS; }
    uint32_t* bin_cap(uint32_t par, uint32_t seg) const { return d_bins + (size_t)(17 + par * 17 + seg) * CRT_RAY_BINS; }
    uint32_t* bin_off(uint32_t par, uint32_t seg) const { return d_bins + (size_t)(17 * 3 + par * 17 + seg) * CRT_RAY_BINS; }
    uint32_t* bin_start(uint32_t seg) const { return d_bins + (size_t)17 * 5 * CRT_RAY_BINS + (size_t)seg * (CRT_RAY_BINS + 1u); }
    uint32_t* bin_ovf(uint32_t seg) const { return d_bins + (size_t)17 * 5 * CRT_RAY_BINS + (size_t)17 * (CRT_RAY_BINS + 1u) + (size_t)seg * 32u; }
    static size_t bins_words() { return (size_t)17 * 5 * CRT_RAY_BINS + (size_t)17 * (CRT_RAY_BINS + 1u) + 17u * 32u; }
    uint32_t debug_fail_batch_alloc = 0;     // test hook (option of the same name): the next growth of the batch buffers fails before it allocates
    uint32_t batch_cap = 1;                  // samples the path state, the ray queues and d_lfinal are sized for (1 until crt_render_frames needs more)
    uint32_t samples_in_stats = 1;           // samples per pixel of the launch the pending stats describe (crt_render_frames batches)
    // event spans behind crt_frame_stats.ms_*: 2 = every launch, 1 = closest-hit launches only, 0 = none (the default: an event-carrying
    // dispatch cannot overlap its neighbours, ~5 us of stream time per launch — 8 % of a Cornell-box frame, 0.8 % of a 1 M-triangle one)
    uint32_t timing = 0;
    bool timing_accumulate = false;          // spans pile up over frames (crt_frame_stats then holds sums) instead of per frame

    // ---- several GPUs behind ONE handle (crt_set_devices): this scene is logical device 0, `peers` are devices 1..n-1, each a
    // complete scene on its own GPU (replicated buffers, own stream, own shard of the tiles).  Every entry point fans out to them;
    // crt_read_sum / crt_resolve gather their packed tile buffers to this device and un-tile the whole frame here.
    std::vector<crt_scene*> peers;           // owned
    crt_scene* primary = nullptr;            // set in a peer
    std::vector<std::pair<size_t, size_t>> scene_bufs;   // (offset of the pointer member, bytes): what a replica needs copied
    uint32_t shard_rank = 0, shard_world = 1; // the caller's shard of the frame (crt_set_shard); rank / world below are this stream's share of it
    uint32_t streams = 1;                    // option "streams": this many tile shards of the frame on streams of their own, on this one GPU
    bool shares_scene = false;               // a replica on its primary's own device: the scene buffers are the primary's, not copies
    std::vector<float*> d_gather;            // per peer, on THIS device: its packed sum buffer as received
    std::vector<uint2*> d_peer_tiles;        // per peer, on THIS device: its local tile list (for the un-tiling launch)
    std::vector<hipEvent_t> ev_peer;         // per peer: "your slice has arrived" (copy transport)
    void* rccl_lib = nullptr;                // dlopen handle; comms[k] = communicator of logical device k
    std::vector<void*> rccl_comms;
    uint32_t gather_transport = 0;           // 0 = RCCL send/recv over xGMI, 1 = hipMemcpyPeerAsync (also: virtual devices, RCCL absent)
    float last_gather_ms = 0.f;

    ~crt_scene() {
        drop_peers();
        hipSetDevice(device);
        if (stream) hipStreamSynchronize(stream);
        if (shares_scene)                    // borrowed from the primary, which frees them
            for (const auto& b : scene_bufs) *reinterpret_cast<void**>(reinterpret_cast<char*>(this) + b.first) = nullptr;
        void* ptrs[] = {d_gamma, d_texcoords, d_textures, d_bvh2, d_tris2, d_nodes, d_planes, d_tris, d_triangles, d_normals, d_materials, d_lights, d_tile_xy, d_sum, d_linear, d_rgba,
                        d_rays[0], d_rays[1], d_nee, d_contrib, d_nee_perm, d_nee_bins, d_qhits, pb.L, pb.T, pb.seed, d_counts,
                        d_t_rays, d_t_hits, d_t_stats, d_visit_totals, d_overflow, d_tile_order, d_tile_cost, d_lfinal, d_bins};
        for (void* p : ptrs) if (p) hipFree(p);
        if (h_tile_cost) hipHostFree(h_tile_cost);
        if (h_tile_order) hipHostFree(h_tile_order);
        if (ev_tile_cost) hipEventDestroy(ev_tile_cost);
        if (ev_tile_order) hipEventDestroy(ev_tile_order);
        if (h_counts) hipHostFree(h_counts);
        if (h_rgba) hipHostFree(h_rgba);
        if (h_visit_totals) hipHostFree(h_visit_totals);
        for (EventSpan& s : spans) { if (s.a) hipEventDestroy(s.a); if (s.b) hipEventDestroy(s.b); }
        if (stream) hipStreamDestroy(stream);
    }

    void drop_peers();      // defined with crt_set_devices

    // grid of a traversal kernel (always a multiple of 8: one slice per XCD group); persistent variant bounded by LDS and registers
    // (the XCD-aware schedule in rt_kernels.hip groups workgroups by blockIdx & 7)
    // reg_cap = workgroups per CU the kernel's VGPR count admits (k_segment ~90 VGPRs -> 5, k_trace/k_shadow <= 64 -> 8)
    // chunk = items one workgroup pass covers: 256 (lock-step kernels) or 1024 (pool kernels)
    uint32_t trace_grid(uint64_t n, uint32_t reg_cap, uint32_t chunk = CRT_TRACE_BLOCK) const {
        // every group's share in one pass: group 0 owns ceil(units / 8) units of 4096 items (== sub_capacity for n = P)
        const uint64_t share = ((n + 4095) / 4096 + 7) / 8 * 4096;
        uint64_t g = 8 * ((share + chunk - 1) / chunk);
        if (oversubscribe != 0u) {
            const uint64_t lds = (uint64_t)(CRT_TRACE_BLOCK / 64) * (stack_entries + CRT_HIT_SLOTS) * 64 * 8;
            const uint64_t per_cu = std::min<uint64_t>(std::min(trace_occupancy, reg_cap), std::max<uint64_t>(1, (160 * 1024) / lds));
            g = std::min(g, ((uint64_t)n_cu * per_cu * oversubscribe + 7) / 8 * 8);
        }
        return (uint32_t)std::max<uint64_t>(g, 8);
    }
    // waves the chip holds at once of a kernel whose wave needs `lds` bytes (handed out in 1,280-byte units) and runs `per_simd` to a SIMD:
    // the grid of a persistent launch (a multiple of 8: one slice per XCD group)
    uint32_t resident_waves(size_t lds, uint32_t per_simd) const {
        const uint64_t units = (lds + 1279) / 1280, by_lds = units ? (160u * 1024u / 1280u) / units : 32u;
        const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(4ull * per_simd, by_lds));
        return (uint32_t)((uint64_t)n_cu * per_cu / 8 * 8);
    }
    uint32_t flat_grid(uint64_t n) const {
        uint64_t blocks = (n + 255) / 256;
        return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(blocks, (uint64_t)n_cu * 8));
    }
    // a span's events are attached to the launches themselves (crt::set_launch_events), not recorded between them
    EventSpan* new_span(int kind) {
        if (timing == 0u || (timing == 1u && kind != 1)) return nullptr;
        if (n_spans >= (int)spans.size()) return nullptr;
        EventSpan* s = &spans[n_spans];
        // the events of a span are created the first time it is used: a scene that never asks for timings creates none (272 event creations
        // were a quarter of a millisecond of every crt_scene_create)
        if (!s->a && hipEventCreate(&s->a) != hipSuccess) { s->a = nullptr; return nullptr; }
        if (!s->b && hipEventCreate(&s->b) != hipSuccess) { s->b = nullptr; return nullptr; }
        ++n_spans;
        s->kind = kind;
        return s;
    }
};

namespace {

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(CRT_ERR_NO_DEVICE, "no HIP device visible: the traversal path has no CPU fallback");
    return CRT_OK;
}

// The tiles of shard `rank` of `world`: the frame's tile x tile squares in Morton order, every world-th one starting at the rank-th
// (SURVEY 8e).  Pure host arithmetic: crt_set_shard, crt_set_devices, option "streams" and the [host] entry crt_shard_tiles share it.
void deal_tiles(uint32_t width, uint32_t height, uint32_t T, uint32_t rank, uint32_t world, std::vector<uint2>& tiles, uint64_t* pixels_in_frame) {
    const uint32_t tx_n = (width + T - 1) / T, ty_n = (height + T - 1) / T;
    struct Item { uint32_t code, x, y; };
    std::vector<Item> all;
    all.reserve((size_t)tx_n * ty_n);
    for (uint32_t y = 0; y < ty_n; ++y)
        for (uint32_t x = 0; x < tx_n; ++x) all.push_back({morton2(x, y), x, y});
    std::sort(all.begin(), all.end(), [](const Item& a, const Item& b) { return a.code < b.code; });
    tiles.clear();
    uint64_t in_frame = 0;
    for (size_t k = rank; k < all.size(); k += world) {
        tiles.push_back(make_uint2(all[k].x, all[k].y));
        const uint32_t w = std::min(T, width - all[k].x * T), h = std::min(T, height - all[k].y * T);
        in_frame += (uint64_t)w * h;
    }
    if (pixels_in_frame) *pixels_in_frame = in_frame;
}

// Logical device k of n_devices, dividing shard base_rank of base_world among themselves (set_devices_of_shard): every n-th tile of that
// shard's list starting at its k-th, i.e. shard base_rank + k * base_world of base_world * n_devices of the whole frame.
inline void device_share(uint32_t k, uint32_t n_devices, uint32_t base_rank, uint32_t base_world, uint32_t* rank, uint32_t* world) {
    *rank = base_rank + k * base_world;
    *world = base_world * n_devices;
}

int build_shard(crt_scene* s) {
    deal_tiles(s->width, s->height, s->tile, s->rank, s->world, s->tiles, &s->n_local_in_frame);
    s->n_local_tiles = (uint32_t)s->tiles.size();
    const uint64_t px = (uint64_t)s->n_local_tiles * s->tile * s->tile;
    if (px >= (1ull << 31)) return fail(CRT_ERR_LIMIT, "framebuffer shard too large for 32-bit pixel indices");
    s->n_local_pixels = (uint32_t)px;
    return CRT_OK;
}

void free_frame_buffers(crt_scene* s) {
    void** ptrs[] = {(void**)&s->d_tile_xy, (void**)&s->d_tile_order, (void**)&s->d_tile_cost, (void**)&s->d_sum, (void**)&s->d_linear, (void**)&s->d_rgba, (void**)&s->d_rays[0],
                     (void**)&s->d_rays[1], (void**)&s->d_nee, (void**)&s->d_contrib, (void**)&s->d_nee_perm, (void**)&s->d_qhits, (void**)&s->pb.L, (void**)&s->pb.T, (void**)&s->pb.seed,
                     (void**)&s->d_lfinal};
    for (void** p : ptrs) { if (*p) hipFree(*p); *p = nullptr; }
    s->batch_cap = 1;
    s->defer_cap = s->defer_regions = s->defer_sub_capacity = s->lfinal_cap = 0;
    // the bins' capacities and offsets describe the queues that were just freed: back to "everything overflows"
    if (s->d_bins) (void)hipMemset(s->d_bins, 0, crt_scene::bins_words() * sizeof(uint32_t));
    if (s->h_tile_cost) { (void)hipHostFree(s->h_tile_cost); s->h_tile_cost = nullptr; }
    if (s->h_tile_order) { (void)hipHostFree(s->h_tile_order); s->h_tile_order = nullptr; }
    s->frame_buffers_ready = false;
}

int alloc_frame_buffers(crt_scene* s) {
    free_frame_buffers(s);
    int rc = build_shard(s);
    if (rc) return rc;
    const size_t P = s->n_local_pixels;
    if ((rc = dev_alloc(&s->d_tile_xy, s->n_local_tiles))) return rc;
    HIPCHK(hipMemcpy(s->d_tile_xy, s->tiles.data(), s->tiles.size() * sizeof(uint2), hipMemcpyHostToDevice));
    {   // processing order: centre-out until a frame has been measured
        const uint32_t nt = s->n_local_tiles, T = s->tile;
        const float cx = 0.5f * (float)((s->width + T - 1) / T) - 0.5f, cy = 0.5f * (float)((s->height + T - 1) / T) - 0.5f;
        std::vector<uint32_t> order(nt);
        for (uint32_t i = 0; i < nt; ++i) order[i] = i;
        auto d2 = [&](uint32_t i) { const float dx = (float)s->tiles[i].x - cx, dy = (float)s->tiles[i].y - cy; return dx * dx + dy * dy; };
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return d2(a) < d2(b); });
        if ((rc = dev_alloc(&s->d_tile_order, std::max<uint32_t>(nt, 1)))) return rc;
        if ((rc = dev_alloc(&s->d_tile_cost, std::max<uint32_t>(nt, 1)))) return rc;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s->h_tile_cost), std::max<uint32_t>(nt, 1) * sizeof(uint32_t)));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s->h_tile_order), std::max<uint32_t>(nt, 1) * sizeof(uint32_t)));
        if (nt) HIPCHK(hipMemcpy(s->d_tile_order, order.data(), nt * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (!s->ev_tile_cost) HIPCHK(hipEventCreateWithFlags(&s->ev_tile_cost, hipEventDisableTiming));
        if (!s->ev_tile_order) HIPCHK(hipEventCreateWithFlags(&s->ev_tile_order, hipEventDisableTiming));
        s->tile_state = crt_scene::TILES_WANT;
        s->tile_order_uploading = false;
        s->tiles_measured_once = false;
        s->tile_cost_spread = 0.f;
    }
    if ((rc = dev_alloc(&s->d_sum, 3 * std::max<size_t>(P, 1)))) return rc;     // a shard may hold no tile at all (more devices or streams than tiles)
    HIPCHK(hipMemset(s->d_sum, 0, 3 * std::max<size_t>(P, 1) * sizeof(float)));
    // a workgroup group handles every 8th unit of 4096 pixels/rays, so it can emit at most this many rays per segment
    s->sub_capacity = (uint32_t)(((P + 4095) / 4096 + 7) / 8 * 4096);
    const size_t Q = 8 * (size_t)s->sub_capacity;
    // the deferred shadow rays' buffers (inplace_shadow 0 / 2) and the hit buffer of the bounce pools (bounce_refill = 1) are allocated
    // by the first frame that needs them
    if (s->max_depth > 1) {                      // path state and ray queues exist only for multi-segment paths
        if ((rc = dev_alloc(&s->d_rays[0], 2 * Q))) return rc;
        if ((rc = dev_alloc(&s->d_rays[1], 2 * Q))) return rc;
        s->rays_doubled = false;
        if ((rc = dev_alloc(&s->pb.L, P))) return rc;
        if ((rc = dev_alloc(&s->pb.T, P))) return rc;
        if ((rc = dev_alloc(&s->pb.seed, P))) return rc;
    }
    s->frame_buffers_ready = true;
    return CRT_OK;
}

crt::FrameArgs frame_args(const crt_scene* s, float rx, float ry) {
    crt::FrameArgs f{};
    f.tile_xy = s->d_tile_xy;
    f.tile_order = s->d_tile_order;
    f.n_local_pixels = s->n_local_pixels;
    f.tile = s->tile; f.width = s->width; f.height = s->height;
    f.tile_log2 = (s->tile & (s->tile - 1u)) == 0u ? (uint32_t)__builtin_ctz(s->tile) : 0u;
    f.jitter = s->jitter;
    f.rv = rx * ry;
    const float W = (float)s->width, H = (float)s->height;
    f.tan_fov = std::tan(s->cam.fov * 0.5f);
    f.aspect_tan = W / H * f.tan_fov;
    for (int k = 0; k < 3; ++k) {
        f.cam_pos[k] = s->cam.position[k]; f.cam_right[k] = s->cam.right[k];
        f.cam_up[k] = s->cam.up[k]; f.cam_forward[k] = s->cam.forward[k];
    }
    return f;
}

int collect_stats(crt_scene* s) {
    if (!s->stats_pending) return CRT_OK;
    HIPCHK(hipStreamSynchronize(s->stream));
    crt_frame_stats st{};
    for (int i = 0; i < s->n_spans; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s->spans[i].a, s->spans[i].b) != hipSuccess) continue;
        switch (s->spans[i].kind) {
            case 0: st.ms_raygen += ms; break;
            case 1: st.ms_trace_closest += ms; st.n_trace_launches++; break;
            case 2: st.ms_trace_any += ms; st.n_trace_launches++; break;
            default: st.ms_shade += ms; break;
        }
    }
    if (s->n_spans > 0) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s->spans[0].a, s->spans[s->n_spans - 1].b) == hipSuccess) st.ms_total = ms;
    }
    st.closest_rays = s->stats.closest_rays;
    st.any_rays = s->stats.any_rays;
    if (s->stats_from_frame && s->stats_counted && s->h_visit_totals) {
        st.nodes_closest = s->h_visit_totals[0];
        st.tris_closest = s->h_visit_totals[1];
        st.nodes_any = s->h_visit_totals[2]; st.tris_any = s->h_visit_totals[3];
        st.wave_steps_closest_nodes = s->h_visit_totals[4]; st.wave_steps_closest_tris = s->h_visit_totals[5];
        st.wave_steps_any_nodes = s->h_visit_totals[6]; st.wave_steps_any_tris = s->h_visit_totals[7];
        st.closest_hits = s->h_visit_totals[8];
        st.nodes_closest_uniform = s->h_visit_totals[10]; st.nodes_any_uniform = s->h_visit_totals[11];
    }
    s->stats = st;
    s->stats_pending = false;
    return CRT_OK;
}

}  // namespace

// device, stream, environment overrides and the descriptor's scalars: shared by both ways of creating a scene
static int init_scene_common(crt_scene* s, const crt_scene_desc* d) {
    if (hipGetDevice(&s->device) != hipSuccess) return (fail(CRT_ERR_HIP, "hipGetDevice failed"));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, s->device) == hipSuccess) s->n_cu = prop.multiProcessorCount;
    if (const char* e = std::getenv("CRT_TIMING")) s->timing = (uint32_t)std::max(0, std::atoi(e));
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) return (fail(CRT_ERR_HIP, "hipStreamCreate failed"));
    s->width = d->width; s->height = d->height; s->max_depth = d->max_depth; s->n_lights = (uint32_t)d->n_lights;
    if (d->n_vertices) {
        for (int k = 0; k < 3; ++k) s->bounds_lo[k] = s->bounds_hi[k] = d->vertices[k];
        for (size_t i = 1; i < d->n_vertices; ++i)
            for (int k = 0; k < 3; ++k) {
                s->bounds_lo[k] = std::min(s->bounds_lo[k], d->vertices[3 * i + k]);
                s->bounds_hi[k] = std::max(s->bounds_hi[k], d->vertices[3 * i + k]);
            }
    }
    for (size_t m = 0; m < d->n_materials; ++m)
        s->special_materials = s->special_materials || d->materials[m].albedo[3] == 1.0f || d->materials[m].albedo[3] == 17.0f;
    return CRT_OK;
}

// What a replica on another GPU needs copied (crt_set_devices): the scene-level device buffers and their sizes.
static void note_buf(crt_scene* s, const void* member, size_t bytes) {
    s->scene_bufs.emplace_back((size_t)(reinterpret_cast<const char*>(member) - reinterpret_cast<const char*>(s)), bytes);
}

// The traversal's copies of nodes and records at the build's stride (CRT_NODE_ROWS / CRT_TRI_ROWS): packed input -> padded copy.
template <typename T>
static int pad_rows(crt_scene* s, T** buf, uint32_t rows_in, uint32_t rows_out, size_t n_items) {
    if (rows_in == rows_out || !*buf) return CRT_OK;
    T* padded = nullptr;
    int rc = dev_alloc(&padded, n_items * rows_out);
    if (rc) return rc;
    crt::launch_restride(*buf, rows_in, padded, rows_out, n_items, s->stream);
    if (hipStreamSynchronize(s->stream) != hipSuccess || hipGetLastError() != hipSuccess) { (void)hipFree(padded); return fail(CRT_ERR_HIP, "crt_scene_create: re-striding failed"); }
    (void)hipFree(*buf);
    *buf = padded;
    return CRT_OK;
}

// The nodes' child planes as floats (uniform node steps, rt_kernels.hip node8_intersect_planes): 192 B per node, 2.4 x the node array — built
// when a frame first walks the CWBVH, so a scene that only ever renders through its BVH2 (option accel 1 / 2) or only serves crt_trace never
// holds them.  In stream order before the frame's kernels: no host wait.  A replica made later gets its copy with the other scene buffers.
static int ensure_planes(crt_scene* s) {
    if (s->d_planes || !s->d_nodes || !s->info.n_nodes8) return CRT_OK;
    int rc;
    if (s->primary && s->shares_scene) {
        // a replica on its primary's own device that was made before any CWBVH frame (the scene rendered through its BVH2 until now): it borrows
        // the primary's planes like every other scene buffer.  Rare path: one host wait for the primary's stream.
        crt_scene* const p = s->primary;
        if ((rc = ensure_planes(p))) return rc;
        HIPCHK(hipStreamSynchronize(p->stream));
        s->d_planes = p->d_planes;
        note_buf(s, &s->d_planes, (size_t)s->info.n_nodes8 * 12 * sizeof(float4));      // borrowed: the destructor lets go of it
        return CRT_OK;
    }
    if ((rc = dev_alloc(&s->d_planes, (size_t)s->info.n_nodes8 * 12))) return rc;
    note_buf(s, &s->d_planes, (size_t)s->info.n_nodes8 * 12 * sizeof(float4));
    crt::launch_expand_planes(s->d_nodes, (uint32_t)CRT_NODE_ROWS, s->d_planes, s->info.n_nodes8, s->stream);
    if (hipGetLastError() != hipSuccess) return fail(CRT_ERR_HIP, "plane expansion failed");
    return CRT_OK;
}

static int finish_scene_setup(crt_scene* s) {
    int rc;
    if (!s->rows_padded) {           // a replica's buffers arrive padded (crt_set_devices copies them as they are)
        if ((rc = pad_rows(s, &s->d_nodes, 5u, (uint32_t)CRT_NODE_ROWS, (size_t)s->info.n_nodes8))) return rc;
        if ((rc = pad_rows(s, &s->d_tris, 3u, (uint32_t)CRT_TRI_ROWS, (size_t)s->info.n_tris8))) return rc;
        s->rows_padded = true;
    }
    // (the float planes of the uniform node steps: ensure_planes, at the first frame that walks the CWBVH)
    if ((rc = dev_alloc(&s->d_overflow, 1))) return rc;
    if (hipMemset(s->d_overflow, 0, sizeof(uint32_t)) != hipSuccess) return fail(CRT_ERR_HIP, "hipMemset failed");
    if ((rc = dev_alloc(&s->d_counts, 2 * kCounters))) return rc;
    s->spans.resize(kMaxEvents);                 // events themselves: created by new_span when a timing option first needs them
    return CRT_OK;
}

extern "C" {

int crt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// HIP loads a code object when one of its kernels is first looked up.  The library has four (traversal kernels; the GPU builders
// with their sort / scan kernels; the CWBVH converter; scene assembly), and the builders' alone costs 12.5 ms on MI355X — more than the
// whole build of a million triangles it then runs (profiles/r04_build_probe.txt: the first crt_scene_create(... CRT_BUILD_SAH) of a
// process that had already rendered took 22.2 ms, every later one 8.4).  So the first scene a process creates on a device starts a
// thread that loads all four while the caller uploads and renders; a build-on-device creation waits for it instead of loading
// the same code objects itself.  crt_warmup() does the same synchronously.
namespace {
struct Warmer {
    std::mutex m;
    std::thread t;
    std::vector<int> started;                      // HIP devices whose code objects are loaded or loading
    std::atomic<int> rc{0};                        // first failure of a background load (crt_warmup reports it; a first use then loads again and fails loudly itself)
    static int load_all() {
        int e;
        if ((e = crt::warm_rt_kernels()) || (e = crt::warm_lbvh_kernels()) || (e = crt::warm_cwbvh_kernels()) || (e = crt::warm_scene_build_kernels())) return e;
        return 0;
    }
    void start(int device) {                       // returns at once
        std::lock_guard<std::mutex> g(m);
        if (device < 0 || std::find(started.begin(), started.end(), device) != started.end()) return;
        if (t.joinable()) t.join();
        try {
            started.push_back(device);
            t = std::thread([this, device] {
                int e = (int)hipSetDevice(device);
                if (!e) e = load_all();
                int none = 0;
                if (e) rc.compare_exchange_strong(none, e);
            });
        } catch (const std::exception&) { /* no thread: the code objects load at first use, as before */ }
    }
    int wait() {                                   // joins the background load; its result (0 = fine)
        std::lock_guard<std::mutex> g(m);
        if (t.joinable()) t.join();
        return rc.load();
    }
    ~Warmer() { if (t.joinable()) t.join(); }
};
Warmer g_warmer;
}  // namespace

// Optional, synchronous: the HIP context of the current device, the library's code objects, and the runtime's own first-use set-up
// (first stream, first host-to-device copy, first event, first kernel dispatch: ~85 ms in a fresh process, profiles/r04_build_probe.txt)
// — what the first crt_scene_create of a process otherwise pays.
int crt_warmup(void) {
    int rc = require_device();
    if (rc) return rc;
    HIPCHK(hipFree(nullptr));                          // creates the context
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    const int bg = g_warmer.wait();
    if (bg) return fail(CRT_ERR_HIP, std::string("crt_warmup: a background load of a code object had failed: ") + hipGetErrorString((hipError_t)bg));
    const int e = Warmer::load_all();
    if (e) return fail(CRT_ERR_HIP, std::string("crt_warmup: loading a code object failed: ") + hipGetErrorString((hipError_t)e));
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr;
    uint4* d = nullptr;
    std::vector<uint4> h(4096, make_uint4(1u, 2u, 3u, 4u));           // 64 KB: through the pageable-copy staging path
    auto done = [&](int code) { if (ev) (void)hipEventDestroy(ev); if (d) (void)hipFree(d); if (st) (void)hipStreamDestroy(st); return code; };
#define W_CHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return done(fail(CRT_ERR_HIP, std::string("crt_warmup: " #expr ": ") + hipGetErrorString(e_))); } while (0)
    W_CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    W_CHK(hipEventCreate(&ev));
    W_CHK(hipMalloc(reinterpret_cast<void**>(&d), 2 * h.size() * sizeof(uint4)));
    W_CHK(hipMemcpyAsync(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice, st));
    W_CHK(hipMemsetAsync(d + h.size(), 0, h.size() * sizeof(uint4), st));
    crt::launch_restride(d, 1u, d + h.size(), 1u, h.size(), st);       // one dispatch of a library kernel
    W_CHK(hipEventRecord(ev, st));
    W_CHK(hipMemcpyAsync(h.data(), d + h.size(), 64, hipMemcpyDeviceToHost, st));
    W_CHK(hipStreamSynchronize(st));
    W_CHK(hipGetLastError());
#undef W_CHK
    return done(CRT_OK);
}

int crt_has_experiments(void) {
#ifdef CRT_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

static int scene_create_impl(const crt_scene_desc* d, crt_scene** out);
static int scene_create_device_built(const crt_scene_desc* d, crt_scene** out);

int crt_scene_create(const crt_scene_desc* d, crt_scene** out) {
    // nothing may unwind through the C ABI: the vectors built during upload can throw bad_alloc / length_error
    try {
        return scene_create_impl(d, out);
    } catch (const std::exception& e) {
        if (out) *out = nullptr;
        return fail(CRT_ERR_NOMEM, std::string("crt_scene_create: ") + e.what());
    }
}

static int scene_create_impl(const crt_scene_desc* d, crt_scene** out) {
    if (!out) return fail(CRT_ERR_INVALID, "crt_scene_create: null out");
    *out = nullptr;
    if (!d) return fail(CRT_ERR_INVALID, "crt_scene_create: null desc");
    if (d->abi_version != CRT_ABI_VERSION) return fail(CRT_ERR_INVALID, "crt_scene_create: abi_version mismatch");
    if (!d->vertices || !d->triangles || d->n_triangles == 0 || d->n_vertices == 0)
        return fail(CRT_ERR_INVALID, "crt_scene_create: vertices/triangles missing");
    if (!d->materials || d->n_materials == 0) return fail(CRT_ERR_INVALID, "crt_scene_create: materials missing");
    if (d->n_lights && !d->lights) return fail(CRT_ERR_INVALID, "crt_scene_create: lights missing");
    if (!d->bvh && !d->bvh8 && !(d->build_flags & CRT_BUILD_LBVH_ON_DEVICE))
        return fail(CRT_ERR_INVALID, "crt_scene_create: neither bvh nor bvh8 given (and build_flags does not ask for a build on the device)");
    if ((d->build_flags & CRT_BUILD_LBVH_ON_DEVICE) && (d->bvh || d->bvh8 || d->tri_orig_ids))
        return fail(CRT_ERR_INVALID, "crt_scene_create: CRT_BUILD_LBVH_ON_DEVICE takes the triangles in source order, without bvh / bvh8 / tri_orig_ids");
    if (d->build_flags & ~(uint32_t)(CRT_BUILD_LBVH_ON_DEVICE | CRT_BUILD_PLOC | CRT_BUILD_SAH | 0xff00u)) return fail(CRT_ERR_INVALID, "crt_scene_create: unknown build_flags");
    if ((d->build_flags & (CRT_BUILD_PLOC | CRT_BUILD_SAH | 0xff00u)) && !(d->build_flags & CRT_BUILD_LBVH_ON_DEVICE))
        return fail(CRT_ERR_INVALID, "crt_scene_create: CRT_BUILD_PLOC / CRT_BUILD_SAH only qualify CRT_BUILD_LBVH_ON_DEVICE");
    if (d->width == 0 || d->height == 0 || d->width > 65535u * 8u || d->height > 65535u * 8u)
        return fail(CRT_ERR_INVALID, "crt_scene_create: bad resolution");
    if (d->max_depth == 0 || d->max_depth > 16) return fail(CRT_ERR_INVALID, "crt_scene_create: max_depth must be 1..16");
    if (d->n_triangles >= (1ull << 31) || d->n_vertices >= (1ull << 31)) return fail(CRT_ERR_LIMIT, "crt_scene_create: too many elements");

    // index validation: a bad index would be an out-of-bounds device access (build-on-device scenes run the same checks
    // as a kernel over the uploaded array, scene_create_device_built)
    const bool host_checks = !(d->build_flags & CRT_BUILD_LBVH_ON_DEVICE);
    for (size_t i = 0; host_checks && i < d->n_triangles; ++i) {
        const crt_triangle& t = d->triangles[i];
        for (int j = 0; j < 3; ++j)
            if (t.v[j] < 0 || (size_t)t.v[j] >= d->n_vertices) return fail(CRT_ERR_INVALID, "crt_scene_create: vertex index out of range");
        if (t.v[3] < 0 || (size_t)t.v[3] >= d->n_materials) return fail(CRT_ERR_INVALID, "crt_scene_create: material index out of range");
        if (t.vn[3] != 0)
            for (int j = 0; j < 3; ++j)
                if (t.vn[j] < 0 || (size_t)t.vn[j] >= d->n_normals || !d->normals)
                    return fail(CRT_ERR_INVALID, "crt_scene_create: normal index out of range");
    }
    // a non-finite vertex turns boxes and areas into NaN (the GPU builders check the same on the device)
    for (size_t i = 0; i < 3 * d->n_vertices; ++i)
        if (!(std::fabs(d->vertices[i]) <= 1.0e18f)) return fail(CRT_ERR_INVALID, "crt_scene_create: a vertex coordinate is not finite or exceeds 1e18");
    const bool have_tex = d->albedo_textures && d->n_textures > 0;
    if (have_tex && (d->tex_width == 0 || d->tex_height == 0 || d->tex_width > 16384 || d->tex_height > 16384))
        return fail(CRT_ERR_INVALID, "crt_scene_create: bad texture size");
    for (size_t m = 0; m < d->n_materials; ++m) {
        const float tx = d->materials[m].tex_ind[0];
        if (have_tex && tx != -1.0f) {
            if (!(tx >= 0.0f && tx < (float)d->n_textures)) return fail(CRT_ERR_INVALID, "crt_scene_create: material texture index out of range");
            if (!d->texcoords) return fail(CRT_ERR_INVALID, "crt_scene_create: textured material but no texcoords");
        }
    }
    if (have_tex && host_checks)
        for (size_t i = 0; i < d->n_triangles; ++i) {
            const crt_triangle& t = d->triangles[i];
            const float tx = d->materials[t.v[3] < 0 || (size_t)t.v[3] >= d->n_materials ? 0 : t.v[3]].tex_ind[0];
            if (tx == -1.0f) continue;
            for (int j = 0; j < 3; ++j)
                if (t.vt[j] < 0 || (size_t)t.vt[j] >= d->n_texcoords) return fail(CRT_ERR_INVALID, "crt_scene_create: texcoord index out of range");
        }
    for (size_t m = 0; m < d->n_materials; ++m) {
        const float ew = d->materials[m].emission[3];
        if (ew != -1.0f && !(ew >= 0.0f && (size_t)ew < d->n_lights))
            return fail(CRT_ERR_INVALID, "crt_scene_create: emissive material refers to a light that does not exist");
    }
    int rc = require_device();
    if (rc) return rc;
    {   // the library's other code objects load in the background while this scene is uploaded (see Warmer)
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) g_warmer.start(dev);
    }
    if (d->build_flags & CRT_BUILD_LBVH_ON_DEVICE) return scene_create_device_built(d, out);

    // CWBVH: take the caller's, or convert the BVH2 (cwbvh.h:58)
    crt::CWBVH conv;
    const crt_node8* nodes8 = d->bvh8;
    size_t n_nodes8 = d->n_bvh8;
    const int32_t* tri_slots = d->bvh8_tri_slots;
    size_t n_tris8 = d->n_bvh8_tris;
    uint32_t depth8 = 0;
    struct HandleGuard { crt_cwbvh* h = nullptr; ~HandleGuard() { if (h) crt_cwbvh_free(h); } } dev_conv;
    if (!nodes8 && d->n_bvh >= 4096) {
        // big trees: the device converter (same bytes as the host one, ~20x faster at 1 M triangles)
        rc = crt_cwbvh_convert_device(d->bvh, d->n_bvh, d->n_triangles, &dev_conv.h);
        if (rc) return fail(rc, std::string("crt_scene_create: BVH2 -> CWBVH failed: ") + crt_last_error());
        nodes8 = crt_cwbvh_nodes(dev_conv.h); n_nodes8 = crt_cwbvh_num_nodes(dev_conv.h);
        tri_slots = crt_cwbvh_tri_slots(dev_conv.h); n_tris8 = crt_cwbvh_num_tris(dev_conv.h);
    } else if (!nodes8) {
        if (!conv.convert(d->bvh, d->n_bvh, d->n_triangles, d->tri_orig_ids))
            return fail(CRT_ERR_INVALID, "crt_scene_create: BVH2 -> CWBVH failed: " + conv.error);
        nodes8 = conv.nodes.data(); n_nodes8 = conv.nodes.size();
        tri_slots = conv.tri_slots.data(); n_tris8 = conv.tri_slots.size();
    } else if (!tri_slots) {
        return fail(CRT_ERR_INVALID, "crt_scene_create: bvh8 given without bvh8_tri_slots");
    }
    {
        std::string why = crt::validate_cwbvh(nodes8, n_nodes8, n_tris8, CRT_STACK_ENTRIES, &depth8);
        if (!why.empty()) return fail(why.find("deeper") != std::string::npos ? CRT_ERR_LIMIT : CRT_ERR_INVALID,
                                      "crt_scene_create: CWBVH rejected: " + why);
        for (size_t i = 0; i < n_tris8; ++i)
            if (tri_slots[i] < 0 || (size_t)tri_slots[i] >= d->n_triangles)
                return fail(CRT_ERR_INVALID, "crt_scene_create: bvh8_tri_slots entry out of range");
    }

    // the traversal loops address node and record rows as base + a 32-bit byte offset (rt_kernels.hip node_rows / tri_rows)
    if (n_nodes8 * (uint64_t)(CRT_NODE_ROWS * 16) >= (1ull << 32) || n_tris8 * (uint64_t)(CRT_TRI_ROWS * 16) >= (1ull << 32))
        return fail(CRT_ERR_LIMIT, "crt_scene_create: CWBVH node or triangle-record array of 4 GiB or more (53 M nodes / 89 M records)");
    std::unique_ptr<crt_scene> owner(new (std::nothrow) crt_scene);   // freed on every early return and on a throw
    crt_scene* s = owner.get();
    if (!s) return fail(CRT_ERR_NOMEM, "crt_scene_create: out of memory");
    auto bail = [&](int code) { return code; };
    if ((rc = init_scene_common(s, d))) return bail(rc);
    // A walk pushes at most one entry per level it has descended FROM — what is left of that node's hit list — and the deepest level (depth8,
    // root = 1) has no inner children to descend to: depth8 - 1 entries hold any walk.  One row of the wave's LDS region is 512 bytes and LDS is
    // handed out in 1,280-byte units: at depth 11 (8 M triangles) the row saved is the difference between 21 and 24 waves per CU.
    s->stack_entries = std::min<uint32_t>(CRT_STACK_ENTRIES, std::max<uint32_t>(2, depth8 - 1u));
    s->info.n_nodes8 = n_nodes8; s->info.n_tris8 = n_tris8; s->info.n_bvh2_nodes = d->n_bvh; s->info.max_depth8 = depth8;

    // pre-gathered intersection records in CWBVH triangle order: (v0|orig id) (e1|slot) (e2|material).
    // e1 = v1 - v0, e2 = v2 - v0 are the subtractions of path_trace.fs:337-338, done once here.
    std::vector<float4> recs(3 * n_tris8);
    for (size_t i = 0; i < n_tris8; ++i) {
        const int32_t slot = tri_slots[i];
        const crt_triangle& t = d->triangles[slot];
        const float* v0 = d->vertices + 3 * (size_t)t.v[0];
        const float* v1 = d->vertices + 3 * (size_t)t.v[1];
        const float* v2 = d->vertices + 3 * (size_t)t.v[2];
        const int32_t id = d->tri_orig_ids ? d->tri_orig_ids[slot] : slot;
        float4 a, b, c;
        a.x = v0[0]; a.y = v0[1]; a.z = v0[2]; std::memcpy(&a.w, &id, 4);
        b.x = v1[0] - v0[0]; b.y = v1[1] - v0[1]; b.z = v1[2] - v0[2]; std::memcpy(&b.w, &slot, 4);
        c.x = v2[0] - v0[0]; c.y = v2[1] - v0[1]; c.z = v2[2] - v0[2]; std::memcpy(&c.w, &t.v[3], 4);
        recs[3 * i] = a; recs[3 * i + 1] = b; recs[3 * i + 2] = c;
    }

#define UP(dst, src, count, T)                                                                             \
    do {                                                                                                   \
        if ((rc = dev_alloc(&(dst), (count)))) return bail(rc);                                            \
        if ((count) && hipMemcpy((dst), (src), (count) * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)  \
            return bail(fail(CRT_ERR_HIP, "hipMemcpy H2D failed"));                                        \
    } while (0)
    static_assert(sizeof(crt_node8) == 80, "node8 must be 80 bytes (cwbvh.h:11-25)");
    static_assert(sizeof(crt_triangle) == 48 && sizeof(crt_flatnode) == 32 && sizeof(crt_material) == 64 && sizeof(crt_light) == 72, "layout");
    UP(s->d_nodes, reinterpret_cast<const uint4*>(nodes8), n_nodes8 * 5, uint4);
    UP(s->d_tris, recs.data(), recs.size(), float4);
    UP(s->d_triangles, reinterpret_cast<const int4*>(d->triangles), d->n_triangles * 3, int4);
    UP(s->d_normals, d->normals, d->n_normals * 3, float);
    UP(s->d_materials, reinterpret_cast<const float4*>(d->materials), d->n_materials * 4, float4);
    UP(s->d_lights, reinterpret_cast<const float*>(d->lights), d->n_lights * 18, float);
    if (have_tex) {
        UP(s->d_texcoords, reinterpret_cast<const float2*>(d->texcoords), d->n_texcoords, float2);
        const size_t n_tex = (size_t)d->tex_width * d->tex_height * d->n_textures * 3;
        std::vector<float> texf(n_tex);
        for (size_t i = 0; i < n_tex; ++i) texf[i] = (float)d->albedo_textures[i] / 255.0f;   // UNORM8 -> float, as the oracle does per fetch
        UP(s->d_textures, texf.data(), n_tex, float);
        s->tex_width = (int32_t)d->tex_width; s->tex_height = (int32_t)d->tex_height; s->n_textures = (int32_t)d->n_textures;
    }
#undef UP
    if (d->bvh) {
        // the BVH2 itself, for the reference-order walk (crt_trace with CRT_TRACE_BVH2): nodes as uploaded, one
        // intersection record per leaf slot, and the stack bound from the tree's depth
        std::vector<uint32_t> level(d->n_bvh, 0);
        uint32_t depth2 = 0;
        for (size_t i = 0; i < d->n_bvh; ++i) {
            if (d->bvh[i].bmax[3] != 0.0f) {
                // leaf: traverse_bvh2 loops over [start, start + range) of the slot-ordered records, so the range must lie
                // inside the triangle array (range <= 255 is the builder's own bit-field cap, FlatNode.h:24-25)
                const float fs = d->bvh[i].bmin[3], fr = d->bvh[i].bmax[3];
                if (!(fr >= 1.0f && fr <= 255.0f) || !(fs >= 0.0f && fs < 16777216.0f) || (size_t)fs + (size_t)fr > d->n_triangles)
                    return bail(fail(CRT_ERR_INVALID, "crt_scene_create: BVH2 leaf range outside the triangle array"));
                depth2 = std::max(depth2, level[i]);
                continue;
            }
            const float fl = d->bvh[i].bmin[3];
            if (!(fl >= 0.0f && fl < 16777216.0f)) return bail(fail(CRT_ERR_INVALID, "crt_scene_create: BVH2 child link out of order"));
            const size_t l = (size_t)fl;
            if (l <= i || l + 1 >= d->n_bvh) return bail(fail(CRT_ERR_INVALID, "crt_scene_create: BVH2 child link out of order"));
            level[l] = level[l + 1] = level[i] + 1;
        }
        if (depth2 + 2 > 96) return bail(fail(CRT_ERR_LIMIT, "crt_scene_create: BVH2 deeper than 94 levels"));
        s->bvh2_stack = depth2 + 2;
        std::vector<float4> rec2(3 * d->n_triangles);
        for (size_t slot = 0; slot < d->n_triangles; ++slot) {
            const crt_triangle& t = d->triangles[slot];
            const float* v0 = d->vertices + 3 * (size_t)t.v[0];
            const float* v1 = d->vertices + 3 * (size_t)t.v[1];
            const float* v2 = d->vertices + 3 * (size_t)t.v[2];
            const int32_t id = d->tri_orig_ids ? d->tri_orig_ids[slot] : (int32_t)slot, sl = (int32_t)slot;
            float4 a, b, c;
            a.x = v0[0]; a.y = v0[1]; a.z = v0[2]; std::memcpy(&a.w, &id, 4);
            b.x = v1[0] - v0[0]; b.y = v1[1] - v0[1]; b.z = v1[2] - v0[2]; std::memcpy(&b.w, &sl, 4);
            c.x = v2[0] - v0[0]; c.y = v2[1] - v0[1]; c.z = v2[2] - v0[2]; std::memcpy(&c.w, &t.v[3], 4);
            rec2[3 * slot] = a; rec2[3 * slot + 1] = b; rec2[3 * slot + 2] = c;
        }
        if ((rc = dev_alloc(&s->d_bvh2, d->n_bvh * 2))) return bail(rc);
        if (hipMemcpy(s->d_bvh2, d->bvh, d->n_bvh * sizeof(crt_flatnode), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(CRT_ERR_HIP, "hipMemcpy H2D failed"));
        if ((rc = dev_alloc(&s->d_tris2, rec2.size()))) return bail(rc);
        if (hipMemcpy(s->d_tris2, rec2.data(), rec2.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(CRT_ERR_HIP, "hipMemcpy H2D failed"));
    }
    note_buf(s, &s->d_nodes, n_nodes8 * (size_t)CRT_NODE_ROWS * 16);
    note_buf(s, &s->d_tris, n_tris8 * (size_t)CRT_TRI_ROWS * 16);
    note_buf(s, &s->d_triangles, d->n_triangles * sizeof(crt_triangle));
    note_buf(s, &s->d_normals, d->n_normals * 3 * sizeof(float));
    note_buf(s, &s->d_materials, d->n_materials * sizeof(crt_material));
    note_buf(s, &s->d_lights, d->n_lights * sizeof(crt_light));
    if (have_tex) {
        note_buf(s, &s->d_texcoords, d->n_texcoords * sizeof(float2));
        note_buf(s, &s->d_textures, (size_t)d->tex_width * d->tex_height * d->n_textures * 3 * sizeof(float));
    }
    if (d->bvh) {
        note_buf(s, &s->d_bvh2, d->n_bvh * sizeof(crt_flatnode));
        note_buf(s, &s->d_tris2, d->n_triangles * 3 * sizeof(float4));
    }
    if ((rc = finish_scene_setup(s))) return bail(rc);
    *out = owner.release();
    return CRT_OK;
}

// CRT_BUILD_LBVH_ON_DEVICE: upload the seven input arrays once, then LBVH -> CWBVH -> leaf-order triangles and intersection
// records without anything leaving HBM (the host-array entry points crt_lbvh_build / crt_cwbvh_convert_device moved 64 MB
// of FlatNodes over PCIe twice at 1 M triangles).  Temporaries of both builders come from one arena allocation.
static int scene_create_device_built(const crt_scene_desc* d, crt_scene** out) {
    const auto t_begin = std::chrono::steady_clock::now();
    // (the FlatNode array of a scene built here never leaves the device: links of 2^24 or more are kept as bit patterns, host/flatnode_link.hpp)
    if (2ull * d->n_triangles >= crt::kMaxLinkBits) return fail(CRT_ERR_LIMIT, "crt_scene_create: more than 2^29 triangles");
    const bool have_tex = d->albedo_textures && d->n_textures > 0;
    std::unique_ptr<crt_scene> owner(new (std::nothrow) crt_scene);
    crt_scene* s = owner.get();
    if (!s) return fail(CRT_ERR_NOMEM, "crt_scene_create: out of memory");
    int rc = init_scene_common(s, d);
    if (rc) return rc;
    const uint32_t n = (uint32_t)d->n_triangles, n2 = 2u * n - 1u;
    hipStream_t st = s->stream;

    crt::DeviceArena arena;                       // input-order triangles + both builders' temporaries; gone when this returns
    auto P = crt::DeviceArena::padded;
    const uint32_t gpu_build_flags = (d->build_flags & CRT_BUILD_SAH) ? (CRT_GPU_BUILD_SAH | (d->build_flags & 0xff00u))
                                     : (d->build_flags & CRT_BUILD_PLOC) ? (CRT_GPU_BUILD_PLOC | (d->build_flags & 0xff00u)) : 0u;
    const size_t tmp_bytes = std::max(crt::lbvh_tmp_bytes(n, gpu_build_flags), crt::cwbvh_tmp_bytes(n2, n));
    hipError_t he = arena.reserve(P((size_t)n * sizeof(crt_triangle)) + P(4) + P((size_t)n * 4) + P((size_t)n * 4) + tmp_bytes);
    if (he != hipSuccess) return fail(CRT_ERR_NOMEM, std::string("crt_scene_create: hipMalloc: ") + hipGetErrorString(he));
    crt_triangle* d_in = arena.take<crt_triangle>(n);
    uint32_t* d_flag = arena.take<uint32_t>(1);
    uint32_t* d_tri_order = arena.take<uint32_t>(n);
    int32_t* d_tri_slots = arena.take<int32_t>(n);
    const size_t persistent_mark = arena.used;

#define UPA(dst, src, count, T)                                                                                      \
    do {                                                                                                             \
        if ((rc = dev_alloc(&(dst), (count)))) return rc;                                                            \
        if ((count) && hipMemcpyAsync((dst), (src), (count) * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess)   \
            return fail(CRT_ERR_HIP, "hipMemcpy H2D failed");                                                        \
    } while (0)
    float* d_verts = nullptr;
    struct Guard { void* p = nullptr; ~Guard() { if (p) (void)hipFree(p); } } verts_guard, nodes_guard;
    UPA(d_verts, d->vertices, d->n_vertices * 3, float);
    verts_guard.p = d_verts;                      // only the builders and the gather kernels read the vertices
    if (hipMemcpyAsync(d_in, d->triangles, (size_t)n * sizeof(crt_triangle), hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(CRT_ERR_HIP, "hipMemcpy H2D failed");
    UPA(s->d_normals, d->normals, d->n_normals * 3, float);
    UPA(s->d_materials, reinterpret_cast<const float4*>(d->materials), d->n_materials * 4, float4);
    UPA(s->d_lights, reinterpret_cast<const float*>(d->lights), d->n_lights * 18, float);
    if (have_tex) {
        UPA(s->d_texcoords, reinterpret_cast<const float2*>(d->texcoords), d->n_texcoords, float2);
        const size_t n_tex = (size_t)d->tex_width * d->tex_height * d->n_textures * 3;
        std::vector<float> texf(n_tex);
        for (size_t i = 0; i < n_tex; ++i) texf[i] = (float)d->albedo_textures[i] / 255.0f;
        if ((rc = dev_alloc(&s->d_textures, n_tex))) return rc;
        if (hipMemcpy(s->d_textures, texf.data(), n_tex * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(CRT_ERR_HIP, "hipMemcpy H2D failed");
        s->tex_width = (int32_t)d->tex_width; s->tex_height = (int32_t)d->tex_height; s->n_textures = (int32_t)d->n_textures;
    }
#undef UPA
    if (hipMemsetAsync(d_flag, 0, 4, st) != hipSuccess) return fail(CRT_ERR_HIP, "hipMemset failed");
    crt::launch_validate_triangles(d_in, n, (uint32_t)d->n_vertices, (uint32_t)d->n_materials, d->normals ? (uint32_t)d->n_normals : 0u,
                                   (uint32_t)d->n_texcoords, reinterpret_cast<const float*>(s->d_materials), have_tex ? 1 : 0, d_flag, st);
    uint32_t flag = 0;
    if (hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(CRT_ERR_HIP, "crt_scene_create: triangle validation failed to run");
    const float upload_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (flag)
        return fail(CRT_ERR_INVALID, std::string("crt_scene_create: ") + ((flag & 1u) ? "vertex" : (flag & 2u) ? "material" : (flag & 4u) ? "normal" : "texcoord") +
                                         " index out of range");

    // BVH2 (kept: it is the FlatNode array the BVH2 frame mode and CRT_TRACE_BVH2 walk) -> CWBVH
    g_warmer.wait();                              // the builders' code objects: loaded by now, or being loaded by that thread
    if ((rc = dev_alloc(&s->d_bvh2, (size_t)n2 * 2))) return rc;
    uint32_t depth2 = 0, n8 = 0, depth8 = 0;
    float lbvh_ms = 0.f, conv_ms = 0.f;
    rc = crt::lbvh_build_on_device(reinterpret_cast<const int32_t*>(d_in), 12, d_verts, n, gpu_build_flags, arena, reinterpret_cast<crt_flatnode*>(s->d_bvh2), d_tri_order,
                                   &depth2, &lbvh_ms, st);
    if (rc) return fail(rc, std::string("crt_scene_create: LBVH build failed: ") + crt_last_error());
    arena.used = persistent_mark;                 // the LBVH temporaries are dead: the converter reuses the space
    crt_node8* d_nodes8 = nullptr;
    rc = crt::cwbvh_convert_on_device(reinterpret_cast<const crt_flatnode*>(s->d_bvh2), n2, n, arena, d_tri_slots, &d_nodes8, nullptr, &n8, &depth8,
                                      &conv_ms, st);
    if (rc) return fail(rc, std::string("crt_scene_create: BVH2 -> CWBVH failed: ") + crt_last_error());
    s->d_nodes = reinterpret_cast<uint4*>(d_nodes8);
    if (depth8 > CRT_STACK_ENTRIES) return fail(CRT_ERR_LIMIT, "crt_scene_create: CWBVH rejected: CWBVH deeper than the traversal stack");

    // leaf-order triangle array (what sbvh.h:130-139 leaves behind), records for both walks
    if ((rc = dev_alloc(&s->d_triangles, (size_t)n * 3))) return rc;
    if ((rc = dev_alloc(&s->d_tris, (size_t)n * 3))) return rc;
    const bool keep_bvh2 = depth2 + 2u <= 96u;    // the BVH2 walk's LDS stack bound (crt_scene_create: "BVH2 deeper than 94 levels")
    if (keep_bvh2 && (rc = dev_alloc(&s->d_tris2, (size_t)n * 3))) return rc;
    crt::launch_gather_slots(d_in, d_tri_order, d_verts, n, reinterpret_cast<crt_triangle*>(s->d_triangles), keep_bvh2 ? s->d_tris2 : nullptr, st);
    crt::launch_gather_records(d_in, d_tri_order, d_tri_slots, d_verts, n, s->d_tris, st);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return fail(CRT_ERR_HIP, "crt_scene_create: scene assembly kernels failed");
    if (!keep_bvh2) { (void)hipFree(s->d_bvh2); s->d_bvh2 = nullptr; }
    s->bvh2_stack = keep_bvh2 ? depth2 + 2u : 0u;
    s->stack_entries = std::min<uint32_t>(CRT_STACK_ENTRIES, std::max<uint32_t>(2, depth8 - 1u));      // as for host-built trees: depth8 - 1 entries hold any walk
    s->info.n_nodes8 = n8; s->info.n_tris8 = n; s->info.n_bvh2_nodes = n2; s->info.max_depth8 = depth8;
    s->info.built_on_device = 1u; s->info.bvh2_depth = depth2;
    s->info.build_upload_ms = upload_ms; s->info.build_lbvh_device_ms = lbvh_ms; s->info.build_convert_device_ms = conv_ms;
    note_buf(s, &s->d_nodes, (size_t)n8 * CRT_NODE_ROWS * 16);
    note_buf(s, &s->d_tris, (size_t)n * CRT_TRI_ROWS * 16);
    note_buf(s, &s->d_triangles, (size_t)n * sizeof(crt_triangle));
    note_buf(s, &s->d_normals, d->n_normals * 3 * sizeof(float));
    note_buf(s, &s->d_materials, d->n_materials * sizeof(crt_material));
    note_buf(s, &s->d_lights, d->n_lights * sizeof(crt_light));
    if (have_tex) {
        note_buf(s, &s->d_texcoords, d->n_texcoords * sizeof(float2));
        note_buf(s, &s->d_textures, (size_t)d->tex_width * d->tex_height * d->n_textures * 3 * sizeof(float));
    }
    if (keep_bvh2) {
        note_buf(s, &s->d_bvh2, (size_t)n2 * sizeof(crt_flatnode));
        note_buf(s, &s->d_tris2, (size_t)n * 3 * sizeof(float4));
    }
    if ((rc = finish_scene_setup(s))) return rc;
    s->info.build_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    *out = owner.release();
    return CRT_OK;
}

int crt_scene_destroy(crt_scene* s) {
    delete s;
    return CRT_OK;
}

int crt_set_camera(crt_scene* s, const crt_camera* cam) {
    if (!s || !cam) return fail(CRT_ERR_INVALID, "crt_set_camera: null argument");
    if (!s->have_camera || std::memcmp(&s->cam, cam, sizeof *cam) != 0) s->tile_state = crt_scene::TILES_WANT;   // a new view: new tile costs
    s->cam = *cam;
    s->have_camera = true;
    for (crt_scene* p : s->peers) { const int rc = crt_set_camera(p, cam); if (rc) return rc; }
    return CRT_OK;
}

static int set_devices_of_shard(crt_scene* s, const int32_t* devices, uint32_t n_devices, uint32_t tile, uint32_t base_rank, uint32_t base_world);

int crt_set_shard(crt_scene* s, uint32_t rank, uint32_t world, uint32_t tile) {
    if (!s) return fail(CRT_ERR_INVALID, "crt_set_shard: null scene");
    if (world == 0 || rank >= world) return fail(CRT_ERR_INVALID, "crt_set_shard: rank/world");
    if (tile < 8 || (tile & 7u) || tile > 1024) return fail(CRT_ERR_INVALID, "crt_set_shard: tile must be a multiple of 8 in 8..1024");
    if (s->streams > 1u) {                   // the caller shards the frame itself now: its shard runs on one stream
        const int32_t one = s->device;
        const int rc = crt_set_devices(s, &one, 1, tile);
        if (rc) return rc;
    }
    if (!s->peers.empty() || s->primary) return fail(CRT_ERR_INVALID, "crt_set_shard: this scene deals its tiles to its own devices (crt_set_devices)");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->rank = rank; s->world = world; s->tile = tile;
    s->shard_rank = rank; s->shard_world = world;
    return alloc_frame_buffers(s);
}

static int ensure_frame(crt_scene* s) {
    if (!s->frame_buffers_ready) return alloc_frame_buffers(s);
    return CRT_OK;
}

int crt_reset(crt_scene* s) {
    if (!s) return fail(CRT_ERR_INVALID, "crt_reset: null scene");
    HIPCHK(hipSetDevice(s->device));
    int rc = ensure_frame(s);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(s->d_sum, 0, 3 * (size_t)std::max<uint32_t>(s->n_local_pixels, 1) * sizeof(float), s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (crt_scene* p : s->peers) { rc = crt_reset(p); if (rc) break; }
    if (!s->peers.empty()) (void)hipSetDevice(s->device);
    return rc;
}

int crt_set_option(crt_scene* s, const char* name, int value) {
    if (!s || !name) return fail(CRT_ERR_INVALID, "crt_set_option: null argument");
    if (!std::strcmp(name, "jitter")) s->jitter = value ? 1u : 0u;
    else if (!std::strcmp(name, "count_visits")) { s->count_visits = value != 0; s->count_batched = value == 2; }
    else if (!std::strcmp(name, "bounce_refill")) s->bounce_refill = value ? 1u : 0u;
    else if (!std::strcmp(name, "refill_pool") || !std::strcmp(name, "shadow_pool")) {
        if (value != 64 && value != 128 && value != 256 && value != 512) return fail(CRT_ERR_INVALID, std::string("crt_set_option: ") + name + " is 64, 128, 256 or 512");
        (name[0] == 'r' ? s->refill_pool : s->shadow_pool) = (uint32_t)value;
    }
    else if (!std::strcmp(name, "shadow_refill_min")) s->shadow_refill_min = (uint32_t)std::min(65, std::max(1, value));
    else if (!std::strcmp(name, "shadow_waves")) s->shadow_waves = (uint32_t)std::min(8, std::max(1, value));
    else if (!std::strcmp(name, "step_hist_mode")) s->step_hist_mode = value ? 1u : 0u;
    else if (!std::strcmp(name, "persistent")) s->persistent = value ? 1u : 0u;
    else if (!std::strcmp(name, "sort_shadow")) s->sort_shadow = value ? 1u : 0u;
#ifdef CRT_EXPERIMENTS
    else if (!std::strcmp(name, "trace_occupancy")) s->trace_occupancy = (uint32_t)std::max(1, value);
    else if (!std::strcmp(name, "oversubscribe")) s->oversubscribe = (uint32_t)std::max(0, value);
    else if (!std::strcmp(name, "waves_per_workgroup")) {
        if (value != 1 && value != 2 && value != 4) return fail(CRT_ERR_INVALID, "crt_set_option: waves_per_workgroup is 1, 2 or 4");
        s->waves_per_workgroup = (uint32_t)value;
    }
#else
    // persistent grids and multi-wave workgroups live in the CRT_EXPERIMENTS build only (make EXPERIMENTS=1); their default values are accepted
    else if (!std::strcmp(name, "trace_occupancy") || !std::strcmp(name, "oversubscribe") || !std::strcmp(name, "waves_per_workgroup")) {
        const bool is_default = !std::strcmp(name, "waves_per_workgroup") ? value == 1 : !std::strcmp(name, "oversubscribe") ? value == 0 : true;
        if (!is_default) return fail(CRT_ERR_INVALID, std::string("crt_set_option: ") + name + " is an experimental variant: this library was built without CRT_EXPERIMENTS");
    }
#endif
    else if (!std::strcmp(name, "streams")) {
        // k tile shards of the frame, each with its own stream, queues and path state, on this one GPU: the segment launches of one
        // shard fill the tails and gaps of the others' (a multi-segment frame is a chain of dependent launches).  The machinery is
        // crt_set_devices' with the same GPU listed k times; the scene buffers are shared, not copied.
        if (value < 0 || value > 4) return fail(CRT_ERR_INVALID, "crt_set_option: streams is 0 (pick for me) or 1..4");
        // 0: what the measurements say — a frame of a few-node scene is bound by the gaps between its launches (3 streams), a multi-segment
        // frame by the tails of its chain of launches (2), a one-segment frame of a large scene by neither (1)
        if (value == 0) value = s->info.n_nodes8 < 64 ? 3 : s->max_depth > 1 ? 2 : 1;
        if (s->primary) return fail(CRT_ERR_INVALID, "crt_set_option: streams is set on the scene, not on a replica");
        if ((uint32_t)value == s->streams) return CRT_OK;
        if (s->streams == 1u && !s->peers.empty())
            return fail(CRT_ERR_INVALID, "crt_set_option: streams is for a scene on one device (crt_set_devices already deals its tiles to several)");
        const std::vector<int32_t> ids((size_t)value, (int32_t)s->device);
        const int rc = set_devices_of_shard(s, ids.data(), (uint32_t)value, s->tile ? s->tile : 16u, s->shard_rank, s->shard_world);
        if (rc) return rc;
        s->streams = (uint32_t)value;
        return CRT_OK;                       // nothing to pass on to the replicas
    }
    else if (!std::strcmp(name, "wave_samples")) s->wave_samples = value < 0 ? 0u : std::min<uint32_t>(3u, (uint32_t)value);
    else if (!std::strcmp(name, "wide_first")) s->wide_first = value < 0 ? 0u : std::min<uint32_t>(2u, (uint32_t)value);
    else if (!std::strcmp(name, "lanes_per_ray")) {
        if (value != 1 && value != 8) return fail(CRT_ERR_INVALID, "crt_set_option: lanes_per_ray is 1 or 8");
        s->lanes_per_ray = (uint32_t)value;
    }
    else if (!std::strcmp(name, "inplace_shadow")) {
        if (value < 0 || value > 3) return fail(CRT_ERR_INVALID, "crt_set_option: inplace_shadow is 1 (in place), 0 (every segment's shadow rays deferred), 2 (bounce segments' deferred) or 3 (pick for me)");
        s->inplace_shadow = (uint32_t)value;
    }
    else if (!std::strcmp(name, "adaptive_tiles")) { s->adaptive_tiles = value ? 1u : 0u; if (value) s->tile_state = crt_scene::TILES_WANT; }
    else if (!std::strcmp(name, "accel")) {
        if (value < 0 || value > 2) return fail(CRT_ERR_INVALID, "crt_set_option: accel is 0 (CWBVH), 1 (BVH2, reference order) or 2 (BVH2, lowest-id ties)");
        if (value != 0 && !s->d_bvh2) return fail(CRT_ERR_INVALID, "crt_set_option: the scene was created without a BVH2 (desc.bvh)");
        if (value != 0 && s->special_materials)
            return fail(CRT_ERR_INVALID, "crt_set_option: the BVH2 frame mode is the shipped shader, which is Lambert only; this scene has Mirror / Disney materials");
        s->accel = (uint32_t)value;
    }
    else if (!std::strcmp(name, "timing")) s->timing = (uint32_t)std::max(0, value);
    else if (!std::strcmp(name, "timing_accumulate")) {
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipStreamSynchronize(s->stream));
        s->timing_accumulate = value != 0;
        s->n_spans = 0;
        const size_t want = value > 0 ? (size_t)std::min(value, 1 << 14) : 0;   // value = launches to make room for
        try { if (s->spans.size() < want) s->spans.resize(want); } catch (const std::exception&) { return fail(CRT_ERR_NOMEM, "crt_set_option: out of host memory (timing spans)"); }
    }
    else if (!std::strcmp(name, "ray_bins")) s->ray_bins = (uint32_t)std::min(5, std::max(0, value));
    else if (!std::strcmp(name, "debug_fail_batch_alloc")) {
        // one injected failure, on ONE device: 1 = this scene's own, k >= 2 = its (k - 1)-th peer (streams / crt_set_devices); 0 disarms all
        if (value >= 2 && (size_t)(value - 2) < s->peers.size()) s->peers[(size_t)(value - 2)]->debug_fail_batch_alloc = 1u;
        else s->debug_fail_batch_alloc = value == 1 ? 1u : 0u;
        if (value == 0) for (crt_scene* p : s->peers) p->debug_fail_batch_alloc = 0u;
        return CRT_OK;
    }
    else if (!std::strcmp(name, "tri_min")) s->tri_min = (uint32_t)std::min(64, std::max(0, value));   // 0: plain per-lane closest-hit loop (what trees of a few nodes get)
    else if (!std::strcmp(name, "refill_min")) s->refill_min = (uint32_t)std::min(65, std::max(1, value));   // 65: never while a lane is busy (lock-step batches)
    else if (!std::strcmp(name, "trace_pool")) {
        if (value != 64 && value != 128 && value != 256) return fail(CRT_ERR_INVALID, "crt_set_option: trace_pool is 64, 128 or 256");
        s->trace_pool = (uint32_t)value;
    }
    else if (!std::strcmp(name, "gather_transport")) {
        if (value != 0 && value != 1) return fail(CRT_ERR_INVALID, "crt_set_option: gather_transport is 0 (RCCL send / recv) or 1 (hipMemcpyPeerAsync)");
        if (value == 0 && !s->peers.empty() && s->rccl_comms.empty())
            return fail(CRT_ERR_INVALID, "crt_set_option: this scene's devices have no RCCL communicators (virtual devices, or librccl.so could not be loaded)");
        s->gather_transport = (uint32_t)value;
        return CRT_OK;
    }
    else return fail(CRT_ERR_INVALID, std::string("crt_set_option: unknown option ") + name);
    for (crt_scene* p : s->peers) { const int rc = crt_set_option(p, name, value); if (rc) { (void)hipSetDevice(s->device); return rc; } }
    if (!s->peers.empty()) (void)hipSetDevice(s->device);      // "timing_accumulate" visits the peers' devices
    return CRT_OK;
}

// One sample per pixel: raygen -> [closest, shade, any, resolve] x max_depth -> accumulate.  n_samples > 1 (a one-segment path
// whose shadow rays are walked in place: nothing is queued between launches): the same launch renders that many samples of
// every pixel one after the other — what n_samples calls would do, bit for bit, without their launch gaps and kernel tails.
static int ensure_batch_buffers(crt_scene* s, uint32_t cap);
static bool uses_ray_bins(const crt_scene* s) {
    // bounce rays binned between the segments (option "ray_bins"): big trees, CWBVH walks, the lock-step segment kernels
    return s->ray_bins != 0u && s->max_depth > 1u && s->info.n_nodes8 >= 64 && s->accel == 0u && !s->bounce_refill;
}
// Segments [first, max_depth) defer their NEE shadow rays (option "inplace_shadow"); first = max_depth: none does.  BVH2 frames walk in place.
static uint32_t first_deferred_segment(const crt_scene* s) {
    const uint32_t mode = s->inplace_shadow == 3u ? (s->info.n_nodes8 >= 64 ? 2u : 1u) : s->inplace_shadow;
    if (s->accel != 0u || mode == 1u) return s->max_depth;
    return mode == 0u ? 0u : std::min(1u, s->max_depth);
}
// d_lfinal for the samples the path buffers are sized for (frames that fold: several samples of multi-segment paths, deferred shadow rays)
static int ensure_lfinal(crt_scene* s) {
    if (s->d_lfinal && s->lfinal_cap >= s->batch_cap) return CRT_OK;
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->d_lfinal) { (void)hipFree(s->d_lfinal); s->d_lfinal = nullptr; }
    const size_t n = (size_t)s->n_local_pixels * s->batch_cap;
    int rc = dev_alloc(&s->d_lfinal, n);
    if (rc) return rc;
    HIPCHK(hipMemset(s->d_lfinal, 0, n * sizeof(float4)));      // pixels outside the frame are never written: they stay zero
    s->lfinal_cap = s->batch_cap;
    return CRT_OK;
}
// The NEE queue and the contribution slots of the deferring segments, for the samples per launch the path buffers are sized for
static int ensure_defer_buffers(crt_scene* s) {
    const uint32_t regions = s->max_depth - first_deferred_segment(s);
    if (regions == 0u) return CRT_OK;
    if (s->d_nee && s->defer_cap == s->batch_cap && s->defer_regions >= regions && s->defer_sub_capacity == s->sub_capacity) return CRT_OK;
    HIPCHK(hipStreamSynchronize(s->stream));
    for (void** p : {(void**)&s->d_nee, (void**)&s->d_contrib, (void**)&s->d_nee_perm}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    s->defer_cap = s->defer_regions = 0;
    const uint64_t slots = (uint64_t)regions * s->n_local_pixels * s->batch_cap;
    if (slots >= (1ull << 32)) return fail(CRT_ERR_LIMIT, "deferred shadow rays: more than 2^32 contribution slots (segments x pixels x samples per launch); use inplace_shadow 1");
    int rc;
    if ((rc = dev_alloc(&s->d_nee, 2 * 8 * (size_t)s->sub_capacity * regions)) || (rc = dev_alloc(&s->d_contrib, (size_t)slots))) return rc;
    if ((uint64_t)8 * s->sub_capacity * regions >= (1ull << 32)) return fail(CRT_ERR_LIMIT, "deferred shadow rays: more than 2^32 queue entries; use inplace_shadow 1");
    if ((rc = dev_alloc(&s->d_nee_perm, 8 * (size_t)s->sub_capacity * regions))) return rc;
    if (!s->d_nee_bins) {
        if ((rc = dev_alloc(&s->d_nee_bins, 2 * 4096 + 9 * kCounterStride))) return rc;
        HIPCHK(hipMemset(s->d_nee_bins, 0, (2 * 4096 + 9 * kCounterStride) * sizeof(uint32_t)));
    }
    s->defer_cap = s->batch_cap; s->defer_regions = regions; s->defer_sub_capacity = s->sub_capacity;
    return CRT_OK;
}
// Everything a batch of n_samples needs ALLOCATED on this device, and nothing enqueued: a scene on several devices or streams prepares
// all of them before the first launch of any, so that an allocation failing on one leaves no device with half a batch in its sums.
static int prepare_batch(crt_scene* s, uint32_t n_samples) {
    HIPCHK(hipSetDevice(s->device));
    int rc = ensure_frame(s);
    if (rc) return rc;
    if (s->n_local_pixels == 0) return CRT_OK;
    const bool batched_paths = n_samples > 1u && s->max_depth > 1u;       // finished paths leave their radiance in d_lfinal
    if (batched_paths && (rc = ensure_batch_buffers(s, n_samples))) return rc;   // grows to the largest batch ever asked for
    if (first_deferred_segment(s) < s->max_depth && ((rc = ensure_lfinal(s)) || (rc = ensure_defer_buffers(s)))) return rc;
    if (s->count_visits && !s->d_visit_totals) {
        if ((rc = dev_alloc(&s->d_visit_totals, 16))) return rc;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s->h_visit_totals), 16 * sizeof(unsigned long long)));
    }
    // the hit buffer of the bounce pools (bounce_refill = 1): allocated by the first frame that needs it
    const size_t Q = 8 * (size_t)s->sub_capacity;
    if (s->bounce_refill && s->max_depth > 1 && !s->d_qhits && (rc = dev_alloc(&s->d_qhits, Q))) return rc;
    if (uses_ray_bins(s)) {
        if (!s->d_bins) {
            if ((rc = dev_alloc(&s->d_bins, crt_scene::bins_words()))) return rc;
            HIPCHK(hipMemsetAsync(s->d_bins, 0, crt_scene::bins_words() * sizeof(uint32_t), s->stream));     // capacities 0: a first launch overflows entirely
        }
        if (!s->rays_doubled) {
            // twice the entries any launch can emit: the bins' places in the first half, the overflow region — in the worst case every
            // ray of a launch, e.g. the first one after the view changed — in the second.  Between two frames the queues hold nothing.
            const uint32_t Qe = 8u * s->sub_capacity;
            HIPCHK(hipStreamSynchronize(s->stream));
            float4 *r0 = nullptr, *r1 = nullptr;
            if ((rc = dev_alloc(&r0, 4 * (size_t)Qe)) || (rc = dev_alloc(&r1, 4 * (size_t)Qe))) { if (r0) (void)hipFree(r0); return rc; }
            (void)hipFree(s->d_rays[0]); (void)hipFree(s->d_rays[1]);
            s->d_rays[0] = r0; s->d_rays[1] = r1;
            s->rays_doubled = true;
        }
    }
    return CRT_OK;
}

static int render_batch_async(crt_scene* s, uint32_t n_samples, const float* rxs, const float* rys) {
    if (!s) return fail(CRT_ERR_INVALID, "crt_render_frame: null scene");
    if (!s->have_camera) return fail(CRT_ERR_INVALID, "crt_render_frame: crt_set_camera was never called");
    int rc = prepare_batch(s, n_samples);            // sets the device; a no-op when render_batch_all has prepared every device already
    if (rc) return rc;
    if (s->n_local_pixels == 0) return CRT_OK;
    const uint32_t P = s->n_local_pixels;
    const bool batched_paths = n_samples > 1u && s->max_depth > 1u;
    const uint32_t first_deferred = first_deferred_segment(s);         // segments from here on leave their shadow rays to k_shadow_deferred
    const bool any_deferred = first_deferred < s->max_depth;
    const bool folds = batched_paths || any_deferred;                   // finished paths leave their radiance in d_lfinal, k_fold_paths adds it to the sum
    const size_t n_paths = (size_t)P * n_samples;
    const float rx = rxs[0], ry = rys[0];
    const crt::FrameArgs f = frame_args(s, rx, ry);
    if (!s->timing_accumulate) s->n_spans = 0;
    if (s->count_visits) HIPCHK(hipMemsetAsync(s->d_visit_totals, 0, 16 * sizeof(unsigned long long), s->stream));
    // counter banks alternate per frame; k_segment<FIRST> clears the other bank for the frame after this one, so a
    // memset is only needed for the very first frame (or after a failed launch left the banks in an unknown state)
    // tile order: adopt a finished measurement, start one if the view is new
    bool measure_tiles = false;
    if (s->adaptive_tiles && s->n_local_tiles > 1 && !s->capturing) {
        const bool arrived = s->tile_state == crt_scene::TILES_PENDING && hipEventQuery(s->ev_tile_cost) == hipSuccess;
        (void)hipGetLastError();                                    // "not ready" is an answer, not an error to report later
        if (arrived) {
            if (s->tile_order_uploading) HIPCHK(hipEventSynchronize(s->ev_tile_order));     // the pinned order buffer is free again
            const uint32_t nt = s->n_local_tiles;
            // Most expensive first.  A unit of work is 4096 pixels = spu tiles (1 for the default 64x64 tile); unit U belongs to
            // workgroup group U & 7 (one group per XCD, no rebalancing between them) and a group renders its units in order.
            // So group g's k-th tile sits in slot ((k / spu) * 8 + g) * spu + k % spu, and the sorted list is dealt to the
            // groups' k-th places in snake order — 0..7, 7..0, ... — which keeps the groups' sums level and every group's
            // own sequence descending.
            std::vector<uint32_t> sorted, row;
            try { sorted.resize(nt); row.reserve(8); } catch (const std::exception&) { return fail(CRT_ERR_NOMEM, "crt_render_frame: out of host memory (tile order)"); }
            for (uint32_t i = 0; i < nt; ++i) sorted[i] = i;
            const uint32_t* cost = s->h_tile_cost;
            std::sort(sorted.begin(), sorted.end(), [cost](uint32_t a, uint32_t b) { return cost[a] > cost[b] || (cost[a] == cost[b] && a < b); });
            const uint32_t tile_px = s->tile * s->tile, spu = tile_px < 4096u && 4096u % tile_px == 0u ? 4096u / tile_px : 1u;
            // how far the expensive tiles stand above the rest (99th percentile over mean): what wave_samples = 2 decides on
            uint64_t cost_sum = 0;
            for (uint32_t i = 0; i < nt; ++i) cost_sum += cost[i];
            s->tile_cost_spread = cost_sum ? (float)((double)cost[sorted[nt / 100u]] * nt / (double)cost_sum) : 0.f;
            uint32_t next = 0;
            for (uint32_t k = 0; next < nt; ++k) {
                row.clear();
                for (uint32_t g = 0; g < 8u; ++g) {
                    const uint64_t slot = ((uint64_t)(k / spu) * 8u + g) * spu + k % spu;
                    if (slot < nt) row.push_back((uint32_t)slot);
                }
                if (k & 1u) std::reverse(row.begin(), row.end());
                for (uint32_t slot : row) s->h_tile_order[slot] = sorted[next++];
            }
            HIPCHK(hipMemcpyAsync(s->d_tile_order, s->h_tile_order, nt * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
            HIPCHK(hipEventRecord(s->ev_tile_order, s->stream));
            s->tile_order_uploading = true;
            s->tile_state = crt_scene::TILES_DONE;
        }
        ++s->frames_since_tile_measure;
        // a camera that moves every frame would otherwise sort the tiles on the host every frame: at most one measurement per 16
        // frames (the first one at once); the counting kernels have another cost profile (measured: a worse order) and do not measure
        if (s->tile_state == crt_scene::TILES_WANT && !s->count_visits && (!s->tiles_measured_once || s->frames_since_tile_measure >= 16u)) {
            s->tiles_measured_once = true;
            s->frames_since_tile_measure = 0;
            HIPCHK(hipMemsetAsync(s->d_tile_cost, 0, s->n_local_tiles * sizeof(uint32_t), s->stream));
            measure_tiles = true;
        }
    }
    s->bank ^= 1u;
    if (s->accel == 0u) { const int prc = ensure_planes(s); if (prc) return prc; }
    if (!s->counts_clean) HIPCHK(hipMemsetAsync(s->d_counts, 0, 2 * kCounters * sizeof(uint32_t), s->stream));
    s->counts_clean = false;
    uint32_t* const cnt = s->counts();

    for (uint32_t b = 0; b < s->max_depth; ++b) {
        crt::SegmentArgs sa{};
        sa.nodes = s->d_nodes; sa.planes = s->d_planes; sa.tris = s->d_tris; sa.triangles = s->d_triangles; sa.normals = s->d_normals;
        sa.materials = s->d_materials; sa.lights = s->d_lights; sa.n_lights = (int32_t)s->n_lights;
        sa.stack_entries = s->stack_entries;
        sa.texcoords = s->d_texcoords; sa.textures = s->d_textures;
        sa.tex_width = s->tex_width; sa.tex_height = s->tex_height; sa.n_textures = s->n_textures;
        sa.f = f;
        sa.sub_capacity = s->sub_capacity;
        const bool bvh2 = s->accel != 0u;
        // trees of a few nodes: plain per-lane closest-hit loop (tri_min = 0) and no bounce pools
        const bool small_tree = s->info.n_nodes8 < 64;
        const bool inplace = b < first_deferred;            // shadow rays walked inside k_segment (else: left to k_shadow_deferred)
        sa.tri_min = small_tree ? 0u : s->tri_min;
        sa.lanes_log2 = s->lanes_per_ray >= 8u ? 3u : s->lanes_per_ray >= 4u ? 2u : s->lanes_per_ray >= 2u ? 1u : 0u;
        sa.nodes2 = s->d_bvh2; sa.tris2 = s->d_tris2; sa.stack_entries2 = s->bvh2_stack; sa.tie = s->accel == 2u ? 1u : 0u;
        const bool bins = uses_ray_bins(s);              // tables and the queues' overflow halves exist (prepare_batch)
        const uint32_t Qe = 8u * s->sub_capacity;        // entries of the bins' half of a queue = what the sub-queue form holds
        if (bins && b + 1 < s->max_depth) {
            crt::RayBins& o = sa.bins_out;
            o.count = s->bin_count(b + 1); o.cap = s->bin_cap(s->bank, b + 1); o.off = s->bin_off(s->bank, b + 1);
            o.ovf_count = s->bin_ovf(b + 1); o.ovf_base = Qe;
            // option "ray_bins": 1 ballots, 2 per-ray atomics, 3 ballots for the first segment's emission and per-ray atomics after, 4 ranking
            // through LDS, 5 ballots for the first segment's emission (a handful of keys per wave) and LDS ranking after
            o.per_lane = s->ray_bins == 2u ? 1u : s->ray_bins == 3u ? (b > 0 ? 1u : 0u) : s->ray_bins == 4u ? 2u : s->ray_bins == 5u ? (b > 0 ? 2u : 0u) : 0u;
            for (int k = 0; k < 3; ++k) {
                const float ext = s->bounds_hi[k] - s->bounds_lo[k];
                o.origin[k] = s->bounds_lo[k];
                o.scale[k] = ext > 0.f ? 8.0f / ext : 0.f;
            }
        }
        if (bins && b > 0) { sa.bin_start = s->bin_start(b); sa.bin_off_in = s->bin_off(s->bank, b); sa.ovf_base_in = Qe; }
        sa.rays_in = s->d_rays[b & 1]; sa.count_in = cnt + counter_index(b, 0, 0);
        sa.rays_next = s->d_rays[(b + 1) & 1]; sa.count_next = cnt + counter_index(b + 1, 0, 0);
        sa.count_shadow = cnt + counter_index(b, 1, 0);
        if (!inplace) {
            const uint32_t r = b - first_deferred;           // this segment's region of the NEE queue and of the slot array
            sa.shadow = s->d_nee + 2 * (size_t)r * 8u * s->sub_capacity;
            sa.contrib = s->d_contrib + (size_t)r * n_paths;
            sa.slot_first = (uint32_t)((size_t)r * n_paths);
            sa.slot_bit = 1u << (8u + b);
        }
        sa.pb = s->pb; sa.sum = s->d_sum;
        sa.last_segment = (b + 1 == s->max_depth) ? 1u : 0u;
        sa.visit_totals = s->d_visit_totals;
        sa.overflow = s->d_overflow;
        if (b == 0) { sa.zero_counts = s->d_counts + (size_t)(s->bank ^ 1u) * kCounters; sa.n_zero = kCounters; }
        sa.l_final = folds ? s->d_lfinal : nullptr;
        sa.tile_cost = (b == 0 && measure_tiles) ? s->d_tile_cost : nullptr;
        sa.n_samples = b == 0 ? n_samples : 1u;
        // how the samples of a batched launch sit on the hardware (option "wave_samples"): 2 = four samples of a 4 x 4 pixel quadrant in
        // the lanes of one wave — the 64 rays of a wave leave a quarter of the area and agree on their nodes like the rays of a frame of
        // twice the resolution (1 M triangles, 1080p: +8.5 % at 4 samples per launch, +12 % at 8; shards gain as well; trees of a few
        // nodes lose 9 %: more, shorter waves) —, 1 = the samples on the waves of a workgroup, 0 = one after the other in each wave
        const bool lanes_ok = (n_samples & 3u) == 0u && !small_tree && !bvh2;
        sa.wave_samples = 0u;
        if (b == 0 && n_samples > 1u) {
            if (lanes_ok && s->wave_samples >= 2u) sa.wave_samples = 2u;
            else if (s->wave_samples != 3u && s->use_wave_samples()) sa.wave_samples = 1u;
        }
        // the first segment's 6-wave build where the launch is bound by throughput, the 5-wave build where its longest waves set its length
        // (with four samples in the lanes of a wave a launch has four times the waves, each a quarter as long, and the measure above —
        // made for samples that follow each other in a wave — flips too early: the 6-wave build wins down to about a million pixels,
        // i.e. while the launch fills the chip's 6,144 wave slots eight times over; an eighth of the 4K frame 0.405 -> 0.396 ms, half a
        // 1080p frame 0.435 -> 0.421, a quarter 0.225 against 0.231 the other way)
        bool wide_auto = !s->bound_by_longest_waves();
        if (sa.wave_samples == 2u)
            wide_auto = (double)(s->n_local_pixels / 16u) * (double)(n_samples / 4u) >= 8.0 * (double)s->n_cu * 4.0 * 6.0;
        sa.wide_first = (b == 0 && (s->wide_first == 2u ? wide_auto : s->wide_first != 0u)) ? 1u : 0u;
        if (b == 0) { s->last_launch_form = (int)sa.wave_samples; s->last_launch_samples = (int)n_samples; }
        for (uint32_t k = 0; k < 8u; ++k) sa.rv_s[k] = k < n_samples ? rxs[k] * rys[k] : 0.f;
        EventSpan* sp = s->new_span(1);
        // option bounce_refill: the closest hits of a bounce segment through lane-refill pools (k_closest_queue), then a shade-only pass
        const bool pretraced = b > 0 && s->bounce_refill && !small_tree && !bvh2 && s->tri_min != 0u && !bins;
        if (pretraced) {
            crt::QueueTraceArgs qa{};
            qa.nodes = s->d_nodes; qa.tris = s->d_tris; qa.rays = sa.rays_in; qa.count = sa.count_in; qa.hits = s->d_qhits;
            qa.stack_entries = s->stack_entries; qa.sub_capacity = s->sub_capacity; qa.refill_min = s->refill_min; qa.tri_min = s->tri_min;
            qa.pool = s->refill_pool; qa.lanes_log2 = sa.lanes_log2;
            qa.persistent = s->persistent; qa.cursors = cnt + cursor_index_closest(b);
            qa.visit_totals = s->d_visit_totals; qa.overflow = s->d_overflow;
            if (sp) crt::set_launch_events(sp->a, nullptr);                 // span = both launches of the segment
            const size_t lds_q = (size_t)(s->stack_entries + CRT_HIT_SLOTS) * 64 * sizeof(uint2);
            crt::launch_closest_queue(qa, s->count_visits, s->persistent ? s->resident_waves(lds_q, 8) : 8u * ((s->sub_capacity + qa.pool - 1u) / qa.pool), s->stream);
            sa.hits_in = s->d_qhits;
            if (sp) crt::set_launch_events(nullptr, sp->b);
        } else if (sp) crt::set_launch_events(sp->a, sp->b);
        // A workgroup renders ONE chunk (CRT_CHUNK_LOOP), so the grid has to cover every chunk a sub-queue can hold: a group's
        // sub-queue gets the rays of that group's units of 4096 pixels — ceil(units / 8) of them — times the samples of the launch.
        // (P * n_samples spread evenly over the 8 groups is fewer chunks than that when the units do not divide by 8.)
        uint32_t grid = s->trace_grid(P, sa.wide_first ? 6 : 5);
        if (b > 0 && batched_paths) grid *= n_samples;
        const int build = crt::launch_segment(sa, b == 0, pretraced, inplace, bvh2, s->special_materials, s->count_visits, grid, s->waves_per_workgroup, s->stream);
        if (b == 0) { s->last_launch_wide = build & 1; s->last_launch_one_pass = (build >> 1) & 1; }
        if (sa.bins_out.count) {
            // fill counts -> the next launch's index space and ray count, and the next frame's capacities (the other parity)
            crt::BinScanArgs ba{};
            ba.count = s->bin_count(b + 1); ba.cap = s->bin_cap(s->bank, b + 1); ba.start = s->bin_start(b + 1); ba.ovf_count = s->bin_ovf(b + 1);
            ba.n_in = cnt + counter_index(b + 1, 0, 0);
            ba.cap_next = s->bin_cap(s->bank ^ 1u, b + 1); ba.off_next = s->bin_off(s->bank ^ 1u, b + 1);
            ba.queue_entries = Qe;
            crt::launch_bin_scan(ba, s->stream);
        }

    }
    if (any_deferred) {
        // ONE any-hit launch for the shadow rays of every deferring segment: full waves from the first step
        crt::ShadowArgs sh{};
        sh.nodes = s->d_nodes; sh.tris = s->d_tris; sh.shadow = s->d_nee; sh.contrib = s->d_contrib;
        sh.count = cnt + counter_index(first_deferred, 1, 0); sh.count_stride = counter_index(1, 1, 0) - counter_index(0, 1, 0);
        sh.stack_entries = s->stack_entries; sh.sub_capacity = s->sub_capacity;
        sh.pool = s->shadow_pool; sh.refill_min = s->shadow_refill_min; sh.tri_min = s->tri_min ? s->tri_min : 1u;
        sh.lanes_log2 = s->lanes_per_ray >= 8u ? 3u : 0u;
        sh.pools_per_region = (s->sub_capacity + sh.pool - 1u) / sh.pool;
        sh.persistent = s->persistent; sh.n_regions = s->max_depth - first_deferred; sh.cursors = cnt + kCursorShadow;
        if (s->sort_shadow && s->info.n_nodes8 >= 64) {
            // sorted by the cell the rays start in: three small launches (histogram, scan, scatter), then a persistent grid over the sorted order
            crt::NeeSortArgs na{};
            na.shadow = s->d_nee; na.count = sh.count; na.count_stride = sh.count_stride; na.sub_capacity = s->sub_capacity; na.n_queues = sh.n_regions * 8u;
            for (int k = 0; k < 3; ++k) {
                const float ext = s->bounds_hi[k] - s->bounds_lo[k];
                na.origin[k] = s->bounds_lo[k];
                na.scale[k] = ext > 0.f ? 16.0f / ext : 0.f;
            }
            na.hist = s->d_nee_bins; na.cursor = s->d_nee_bins + 4096; na.perm = s->d_nee_perm; na.meta = s->d_nee_bins + 2 * 4096;
            crt::launch_nee_sort(na, (uint32_t)s->n_cu * 2u, s->stream);
            sh.perm = s->d_nee_perm; sh.sort_meta = na.meta; sh.persistent = 1u;
        }
        sh.visit_totals = s->d_visit_totals ? s->d_visit_totals + 2 : nullptr;
        sh.overflow = s->d_overflow;
        EventSpan* sp = s->new_span(2);
        if (sp) crt::set_launch_events(sp->a, sp->b);
        const size_t lds_q = (size_t)(s->stack_entries + CRT_HIT_SLOTS) * 64 * sizeof(uint2);
        crt::launch_shadow_deferred(sh, s->count_visits, sh.persistent ? s->resident_waves(lds_q, s->shadow_waves) : 8u * sh.n_regions * sh.pools_per_region, s->stream);
    }
    if (folds) crt::launch_fold_paths(s->d_sum, s->d_lfinal, any_deferred ? s->d_contrib : nullptr, P, n_samples, first_deferred, s->stream);
    if (s->count_visits)
        HIPCHK(hipMemcpyAsync(s->h_visit_totals, s->d_visit_totals, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    if (measure_tiles) {
        HIPCHK(hipMemcpyAsync(s->h_tile_cost, s->d_tile_cost, s->n_local_tiles * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipEventRecord(s->ev_tile_cost, s->stream));
        s->tile_state = crt_scene::TILES_PENDING;
    }
    s->stats_counted = s->count_visits;
    HIPCHK(hipGetLastError());
    s->counts_clean = true;
    s->stats_pending = true;
    s->stats_from_frame = true;   // ray counts come from h_counts at the next sync
    s->samples_in_stats = n_samples;
    return CRT_OK;
}

// Path state, ray queues and the final-radiance buffer for `cap` samples per launch (multi-segment paths).  The per-group queue
// capacity grows with it, so the lazily allocated shadow queue / hit buffer are dropped and come back at the new size.
static int ensure_batch_buffers(crt_scene* s, uint32_t cap) {
    if (s->batch_cap >= cap) return CRT_OK;
    HIPCHK(hipStreamSynchronize(s->stream));
    // The lazily allocated hit buffer and the deferred shadow rays' buffers follow the per-group capacity: dropped now (null pointers the
    // next frame re-allocates at the size then in force), whatever happens below.
    for (void** p : {(void**)&s->d_nee, (void**)&s->d_contrib, (void**)&s->d_nee_perm, (void**)&s->d_qhits}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    s->defer_cap = s->defer_regions = 0;
    // Everything else is allocated at the new size FIRST and swapped in only when all of it exists: a failed growth (~250 B per
    // pixel and sample: 16 GB for a 4K frame at 8 samples) leaves the scene exactly as it was, able to render frame by frame.
    const size_t P = s->n_local_pixels;
    const uint32_t sub_capacity = (uint32_t)(((P + 4095) / 4096 + 7) / 8 * 4096) * cap;
    const size_t Q = 8 * (size_t)sub_capacity;
    float4 *rays0 = nullptr, *rays1 = nullptr, *L = nullptr, *T = nullptr, *lfinal = nullptr;
    float2* seed = nullptr;
    auto undo = [&](int code) {
        for (void* p : {(void*)rays0, (void*)rays1, (void*)L, (void*)T, (void*)seed, (void*)lfinal}) if (p) (void)hipFree(p);
        return code;
    };
    int rc;
    if (s->debug_fail_batch_alloc) { s->debug_fail_batch_alloc = 0; return fail(CRT_ERR_NOMEM, "hipMalloc: injected failure (debug_fail_batch_alloc)"); }
    if ((rc = dev_alloc(&rays0, 2 * Q)) || (rc = dev_alloc(&rays1, 2 * Q)) || (rc = dev_alloc(&L, P * cap)) || (rc = dev_alloc(&T, P * cap)) ||
        (rc = dev_alloc(&seed, P * cap)) || (rc = dev_alloc(&lfinal, P * cap)))
        return undo(rc);
    if (hipMemset(lfinal, 0, P * cap * sizeof(float4)) != hipSuccess)      // pixels outside the frame are never written: they stay zero
        return undo(fail(CRT_ERR_HIP, "hipMemset failed"));
    for (void* p : {(void*)s->d_rays[0], (void*)s->d_rays[1], (void*)s->pb.L, (void*)s->pb.T, (void*)s->pb.seed, (void*)s->d_lfinal}) if (p) (void)hipFree(p);
    s->d_rays[0] = rays0; s->d_rays[1] = rays1; s->pb.L = L; s->pb.T = T; s->pb.seed = seed; s->d_lfinal = lfinal;
    s->sub_capacity = sub_capacity;
    s->batch_cap = cap;
    s->lfinal_cap = cap;
    s->rays_doubled = false;
    if (s->d_bins) HIPCHK(hipMemset(s->d_bins, 0, crt_scene::bins_words() * sizeof(uint32_t)));     // new queues: the bins start empty-handed again
    return CRT_OK;
}

// how many samples one launch may render: more than one only when nothing travels between launches (one path segment, shadow
// rays walked in place) — bounce queues and the shadow queue hold one entry per pixel
static uint32_t batch_limit(const crt_scene* s) {
    // the batched builds of the first segment walk its shadow rays in place; counting frames run one by one
    if (first_deferred_segment(s) == 0u || (s->count_visits && !s->count_batched)) return 1u;
    if (s->accel != 0u) return 1u;                                     // the BVH2 frame mode (a comparison aid) has no batched build
    if (s->count_visits) {
        // counting in the timed form: exactly the launch of four samples in the lanes of a wave has a counting build (launch_segment)
        const bool lanes_form = s->wave_samples >= 2u && s->info.n_nodes8 >= 64 && s->tri_min != 0u && s->lanes_per_ray >= 8u;
        const uint64_t fit4 = s->n_local_pixels ? 0x7fffffffull / s->n_local_pixels : 4ull;
        return lanes_form && fit4 >= 4ull ? 4u : 1u;
    }
    if (s->max_depth == 1u) return 8u;
    // several segments: every sample keeps its own path state and queue entries (ensure_batch_buffers) and the samples' radiance is
    // added in frame order by k_fold_paths
    // path ids are sample * n_local_pixels + pixel and must stay below 2^31 (bit 31 tags queue entries)
    const uint64_t fit = s->n_local_pixels ? 0x7fffffffull / s->n_local_pixels : 8ull;
    return (uint32_t)std::min<uint64_t>(8ull, std::max<uint64_t>(1ull, fit));         // 1 M triangles, 4 segments: 1.78 / 1.61 / 1.50 / 1.45 ms per frame at 1 / 2 / 4 / 8 frames per launch
}

// one batch on every device of the scene: each enqueues on its own stream and returns, so the devices render concurrently
static int render_batch_all(crt_scene* s, uint32_t n_samples, const float* rxs, const float* rys) {
    int rc = CRT_OK;
    if (!s->peers.empty()) {
        // phase 1: every device's buffers for this batch exist before any device renders (a failed growth leaves every sum as it was)
        if (!s->have_camera) return fail(CRT_ERR_INVALID, "crt_render_frame: crt_set_camera was never called");
        if ((rc = prepare_batch(s, n_samples))) { (void)hipSetDevice(s->device); return rc; }
        for (crt_scene* p : s->peers) if ((rc = prepare_batch(p, n_samples))) { (void)hipSetDevice(s->device); return rc; }
    }
    // phase 2: enqueue everywhere
    rc = render_batch_async(s, n_samples, rxs, rys);
    for (size_t k = 0; !rc && k < s->peers.size(); ++k) rc = render_batch_async(s->peers[k], n_samples, rxs, rys);
    if (!s->peers.empty()) (void)hipSetDevice(s->device);      // the caller's thread keeps the device the scene lives on
    return rc;
}

int crt_render_frame_async(crt_scene* s, float rx, float ry) {
    if (!s) return fail(CRT_ERR_INVALID, "crt_render_frame: null scene");
    return render_batch_all(s, 1u, &rx, &ry);
}

int crt_render_frames_async(crt_scene* s, uint32_t n, const float* rx, const float* ry) {
    if (!s) return fail(CRT_ERR_INVALID, "crt_render_frames: null scene");
    if (n && (!rx || !ry)) return fail(CRT_ERR_INVALID, "crt_render_frames: null argument");
    const uint32_t lim = batch_limit(s);
    const bool fours = s->wave_samples >= 2u && s->info.n_nodes8 >= 64 && s->accel == 0u;     // launches that can put 4 samples in the lanes of a wave
    for (uint32_t i = 0; i < n;) {
        uint32_t k = std::min(lim, n - i);
        if (fours && k > 4u && (k & 3u)) k &= ~3u;                       // 7 frames = 4 + 3, not 7 one after the other
        if (s->count_visits && k != 4u) k = 1u;                          // counting launches: four samples in the lanes form, or one
        const int rc = render_batch_all(s, k, rx + i, ry + i);
        if (rc) return rc;
        i += k;
    }
    return CRT_OK;
}

int crt_render_frames(crt_scene* s, uint32_t n, const float* rx, const float* ry) {
    const int rc = crt_render_frames_async(s, n, rx, ry);
    if (rc) return rc;
    return crt_sync(s);
}

int crt_sync(crt_scene* s) {
    if (!s) return fail(CRT_ERR_INVALID, "crt_sync: null scene");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipGetLastError());
    for (crt_scene* p : s->peers) { const int rc = crt_sync(p); if (rc) return rc; }
    if (!s->peers.empty()) HIPCHK(hipSetDevice(s->device));
    return CRT_OK;
}

int crt_render_frame(crt_scene* s, float rx, float ry) {
    int rc = crt_render_frame_async(s, rx, ry);
    if (rc) return rc;
    return crt_sync(s);
}

int crt_get_frame_stats(crt_scene* s, crt_frame_stats* out) {
    if (!s || !out) return fail(CRT_ERR_INVALID, "crt_get_frame_stats: null argument");
    HIPCHK(hipSetDevice(s->device));
    if (s->stats_pending) {
        HIPCHK(hipStreamSynchronize(s->stream));
        if (s->stats_from_frame) {
            // the queue counters stay valid until the next frame's memset: fetch them only when asked
            if (!s->h_counts && hipHostMalloc(reinterpret_cast<void**>(&s->h_counts), kCounters * sizeof(uint32_t)) != hipSuccess)
                return fail(CRT_ERR_NOMEM, "hipHostMalloc failed");
            HIPCHK(hipMemcpy(s->h_counts, s->counts(), kCounters * sizeof(uint32_t), hipMemcpyDeviceToHost));
            uint64_t closest = 0, any = 0;
            closest = (uint64_t)s->n_local_in_frame * s->samples_in_stats;   // segment 0: one primary ray per in-frame pixel and sample
            for (uint32_t b = 0; b < s->max_depth; ++b)
                for (uint32_t g = 0; g < 8; ++g) {
                    if (b) closest += s->h_counts[counter_index(b, 0, g)];
                    any += s->h_counts[counter_index(b, 1, g)];
                }
            s->stats.closest_rays = closest; s->stats.any_rays = any;
        }
        int rc = collect_stats(s);
        if (rc) return rc;
        HIPCHK(hipMemcpy(&s->stats.stack_overflows, s->d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    *out = s->stats;
    // several devices: ray and visit counts are sums over the devices, times the slowest device's (they run side by side)
    for (crt_scene* p : s->peers) {
        crt_frame_stats ps{};
        const int rc = crt_get_frame_stats(p, &ps);
        if (rc) return rc;
        out->closest_rays += ps.closest_rays; out->any_rays += ps.any_rays;
        out->nodes_closest += ps.nodes_closest; out->tris_closest += ps.tris_closest; out->nodes_any += ps.nodes_any; out->tris_any += ps.tris_any;
        out->wave_steps_closest_nodes += ps.wave_steps_closest_nodes; out->wave_steps_closest_tris += ps.wave_steps_closest_tris;
        out->wave_steps_any_nodes += ps.wave_steps_any_nodes; out->wave_steps_any_tris += ps.wave_steps_any_tris;
        out->closest_hits += ps.closest_hits; out->stack_overflows += ps.stack_overflows;
        out->nodes_closest_uniform += ps.nodes_closest_uniform; out->nodes_any_uniform += ps.nodes_any_uniform;
        out->ms_total = std::max(out->ms_total, ps.ms_total); out->ms_trace_closest = std::max(out->ms_trace_closest, ps.ms_trace_closest);
        out->ms_trace_any = std::max(out->ms_trace_any, ps.ms_trace_any); out->ms_shade = std::max(out->ms_shade, ps.ms_shade);
    }
    if (!s->peers.empty()) HIPCHK(hipSetDevice(s->device));
    return CRT_OK;
}

int crt_get_bvh_info(crt_scene* s, crt_bvh_info* out) {
    if (!s || !out) return fail(CRT_ERR_INVALID, "crt_get_bvh_info: null argument");
    *out = s->info;
    return CRT_OK;
}

// Option "streams": the caller's shard lives in k parts, one per stream; tile i of stream j is tile i * k + j of the shard's own list
// (set_devices_of_shard), so the shard's packed buffer is the parts interleaved tile by tile.
static size_t shard_tiles_of_streams(const crt_scene* s) {
    size_t n = s->n_local_tiles;
    for (const crt_scene* p : s->peers) n += p->n_local_tiles;
    return n;
}
static int copy_packed_of_streams(crt_scene* s, void* dst, hipMemcpyKind kind) {
    const size_t tile_bytes = 3 * (size_t)s->tile * s->tile * sizeof(float);
    const size_t k = s->peers.size() + 1;
    for (crt_scene* p : s->peers) HIPCHK(hipStreamSynchronize(p->stream));     // their frames are in their sums before these are read
    for (size_t j = 0; j < k; ++j) {
        const crt_scene* q = j == 0 ? s : s->peers[j - 1];
        if (!q->n_local_tiles) continue;
        HIPCHK(hipMemcpy2DAsync(static_cast<char*>(dst) + j * tile_bytes, k * tile_bytes, q->d_sum, tile_bytes, tile_bytes, q->n_local_tiles, kind, s->stream));
    }
    return CRT_OK;
}

int crt_packed_info(crt_scene* s, uint32_t* n_local_tiles, uint32_t* tile, size_t* n_floats) {
    if (!s) return fail(CRT_ERR_INVALID, "crt_packed_info: null scene");
    HIPCHK(hipSetDevice(s->device));
    int rc = ensure_frame(s);
    if (rc) return rc;
    // option "streams": the caller's shard is what its streams hold together
    const size_t tiles = s->streams > 1u ? shard_tiles_of_streams(s) : s->n_local_tiles;
    if (n_local_tiles) *n_local_tiles = (uint32_t)tiles;
    if (tile) *tile = s->tile;
    if (n_floats) *n_floats = 3 * tiles * s->tile * s->tile;
    return CRT_OK;
}

int crt_read_packed(crt_scene* s, float* dst, size_t n_floats) {
    if (!s || !dst) return fail(CRT_ERR_INVALID, "crt_read_packed: null argument");
    HIPCHK(hipSetDevice(s->device));
    int rc = ensure_frame(s);
    if (rc) return rc;
    if (s->streams > 1u) {
        if (n_floats != 3 * shard_tiles_of_streams(s) * s->tile * s->tile) return fail(CRT_ERR_INVALID, "crt_read_packed: size mismatch");
        if ((rc = copy_packed_of_streams(s, dst, hipMemcpyDeviceToHost))) return rc;
        HIPCHK(hipStreamSynchronize(s->stream));
        return CRT_OK;
    }
    if (n_floats != 3 * (size_t)s->n_local_pixels) return fail(CRT_ERR_INVALID, "crt_read_packed: size mismatch");
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(dst, s->d_sum, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_copy_packed_device(crt_scene* s, void* d_dst, size_t n_floats, int sync) {
    if (!s || !d_dst) return fail(CRT_ERR_INVALID, "crt_copy_packed_device: null argument");
    HIPCHK(hipSetDevice(s->device));
    int rc = ensure_frame(s);
    if (rc) return rc;
    if (s->streams > 1u) {
        if (n_floats != 3 * shard_tiles_of_streams(s) * s->tile * s->tile) return fail(CRT_ERR_INVALID, "crt_copy_packed_device: size mismatch");
        if ((rc = copy_packed_of_streams(s, d_dst, hipMemcpyDeviceToDevice))) return rc;
        if (sync) HIPCHK(hipStreamSynchronize(s->stream));
        return CRT_OK;
    }
    if (n_floats != 3 * (size_t)s->n_local_pixels) return fail(CRT_ERR_INVALID, "crt_copy_packed_device: size mismatch");
    HIPCHK(hipMemcpyAsync(d_dst, s->d_sum, n_floats * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    if (sync) HIPCHK(hipStreamSynchronize(s->stream));
    return CRT_OK;
}

static int gather_peers(crt_scene* s);

static int untile_to_linear(crt_scene* s) {
    const size_t n = 3 * (size_t)s->width * s->height;
    if (!s->d_linear) { int rc = dev_alloc(&s->d_linear, n); if (rc) return rc; }
    HIPCHK(hipMemsetAsync(s->d_linear, 0, n * sizeof(float), s->stream));
    if (s->n_local_pixels) {
        const crt::FrameArgs f = frame_args(s, 0.f, 0.f);
        crt::launch_untile(f, s->d_sum, s->d_linear, s->flat_grid(s->n_local_pixels), s->stream);
    }
    if (!s->peers.empty()) {
        // the other devices' packed tile buffers come to this device ("multi-GPU gather happens inside these"), then
        // every slice is un-tiled into the one linear frame with its own tile list; tiles are disjoint, so the order does not matter
        int rc = gather_peers(s);
        if (rc) return rc;
        for (size_t k = 0; k < s->peers.size(); ++k) {
            const crt_scene* p = s->peers[k];
            if (!p->n_local_pixels) continue;
            crt::FrameArgs f = frame_args(p, 0.f, 0.f);
            f.tile_xy = s->d_peer_tiles[k];
            f.tile_order = nullptr;
            crt::launch_untile(f, s->d_gather[k], s->d_linear, s->flat_grid(p->n_local_pixels), s->stream);
        }
    }
    return CRT_OK;
}

int crt_read_sum(crt_scene* s, float* rgb, size_t n_floats) {
    if (!s || !rgb) return fail(CRT_ERR_INVALID, "crt_read_sum: null argument");
    if (n_floats != 3 * (size_t)s->width * s->height) return fail(CRT_ERR_INVALID, "crt_read_sum: n_floats must be width*height*3");
    HIPCHK(hipSetDevice(s->device));
    int rc = ensure_frame(s);
    if (rc) return rc;
    if ((rc = untile_to_linear(s))) return rc;
    HIPCHK(hipMemcpyAsync(rgb, s->d_linear, n_floats * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CRT_OK;
}

int crt_sum_device(crt_scene* s, const float** d_rgb) {
    if (!s || !d_rgb) return fail(CRT_ERR_INVALID, "crt_sum_device: null argument");
    *d_rgb = nullptr;
    HIPCHK(hipSetDevice(s->device));
    int rc = ensure_frame(s);
    if (rc) return rc;
    if ((rc = untile_to_linear(s))) return rc;
    HIPCHK(hipStreamSynchronize(s->stream));
    *d_rgb = s->d_linear;
    return CRT_OK;
}

// The pinned gamma of the resolve kernels: thr[j] (j = 1..255) = the smallest float x >= 0 whose reference byte
// (uint8)(clamp01((float)pow((double)x, 1 / 2.2)) * 255 + 0.5) is >= j — pow in double, rounded once.  The byte of any x is then the number of
// thresholds it has reached, whatever the device's own powf does in its last bits.  (oracle/oracle.c builds the same table by itself.)
static const float* gamma_thresholds() {
    static float thr[256];
    static std::once_flag once;
    std::call_once(once, [] {
        auto ref_byte = [](float x) {
            float v = (float)std::pow((double)x, (double)(1.0f / 2.2f));
            v = v < 0.f ? 0.f : v > 1.f ? 1.f : v;
            return (uint32_t)(uint8_t)(v * 255.0f + 0.5f);
        };
        thr[0] = 0.f;
        for (uint32_t j = 1; j < 256u; ++j) {
            uint32_t lo = 0u, hi = 0x7f800000u;              // bit patterns of the non-negative floats, ordered like their values; ref_byte(+inf) = 255
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2u;
                float x; std::memcpy(&x, &mid, 4);
                if (ref_byte(x) >= j) hi = mid; else lo = mid + 1u;
            }
            std::memcpy(&thr[j], &lo, 4);
        }
    });
    return thr;
}

// tone-mapped RGBA8 of the frame in s->d_rgba (this device), straight from the packed tile buffers: this scene's, and its peers' as gathered
static int resolve_to_device(crt_scene* s, float inv_count) {
    const size_t npx = (size_t)s->width * s->height;
    int rc;
    if (!s->d_rgba) { if ((rc = dev_alloc(&s->d_rgba, 4 * npx))) return rc; }
    if (!s->d_gamma) {
        if ((rc = dev_alloc(&s->d_gamma, 256))) return rc;
        HIPCHK(hipMemcpy(s->d_gamma, gamma_thresholds(), 256 * sizeof(float), hipMemcpyHostToDevice));
    }
    // a shard of a frame (crt_set_shard with world > 1) leaves the other ranks' pixels at 0
    if (s->shard_world > 1u) HIPCHK(hipMemsetAsync(s->d_rgba, 0, 4 * npx, s->stream));
    if (s->n_local_pixels) {
        const crt::FrameArgs f = frame_args(s, 0.f, 0.f);
        crt::launch_resolve_packed(f, s->d_sum, inv_count, s->d_gamma, s->d_rgba, s->flat_grid(s->n_local_pixels), s->stream);
    }
    if (!s->peers.empty()) {
        if ((rc = gather_peers(s))) return rc;
        for (size_t k = 0; k < s->peers.size(); ++k) {
            const crt_scene* p = s->peers[k];
            if (!p->n_local_pixels) continue;
            crt::FrameArgs f = frame_args(p, 0.f, 0.f);
            f.tile_xy = s->d_peer_tiles[k];
            f.tile_order = nullptr;
            crt::launch_resolve_packed(f, s->d_gather[k], inv_count, s->d_gamma, s->d_rgba, s->flat_grid(p->n_local_pixels), s->stream);
        }
    }
    return CRT_OK;
}

int crt_resolve(crt_scene* s, float inv_count, uint8_t* rgba, size_t n_bytes) {
    if (!s || !rgba) return fail(CRT_ERR_INVALID, "crt_resolve: null argument");
    const size_t npx = (size_t)s->width * s->height;
    if (n_bytes != 4 * npx) return fail(CRT_ERR_INVALID, "crt_resolve: n_bytes must be width*height*4");
    HIPCHK(hipSetDevice(s->device));
    int rc = ensure_frame(s);
    if (rc) return rc;
    if ((rc = resolve_to_device(s, inv_count))) return rc;
    // through a pinned staging buffer: a copy into pageable memory is staged by the runtime in small pieces (8.3 MB: ~2.5 ms against 0.4)
    if (!s->h_rgba) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s->h_rgba), 4 * npx));
    HIPCHK(hipMemcpyAsync(s->h_rgba, s->d_rgba, n_bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    std::memcpy(rgba, s->h_rgba, n_bytes);
    return CRT_OK;
}

int crt_resolve_device(crt_scene* s, float inv_count, const uint8_t** d_rgba, int sync) {
    if (!s || !d_rgba) return fail(CRT_ERR_INVALID, "crt_resolve_device: null argument");
    *d_rgba = nullptr;
    HIPCHK(hipSetDevice(s->device));
    int rc = ensure_frame(s);
    if (rc) return rc;
    if ((rc = resolve_to_device(s, inv_count))) return rc;
    if (sync) HIPCHK(hipStreamSynchronize(s->stream));
    *d_rgba = s->d_rgba;
    return CRT_OK;
}

int crt_get_launch_times(crt_scene* s, float* ms, size_t cap, size_t* n_out) {
    if (!s || !n_out) return fail(CRT_ERR_INVALID, "crt_get_launch_times: null argument");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    size_t n = 0;
    for (int i = 0; i < s->n_spans; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, s->spans[i].a, s->spans[i].b) != hipSuccess) continue;
        if (ms && n < cap) ms[n] = t;
        ++n;
    }
    *n_out = n;
    return CRT_OK;
}

// ------------------------------------------------------------------ several GPUs, one handle ----

namespace {

// The slice of RCCL this file uses, resolved from librccl.so at crt_set_devices time: the library is a dependency only of scenes
// that span several GPUs (and a process that already carries an RCCL — PyTorch ships its own — keeps exactly that one).
struct Rccl {
    typedef int (*InitAll)(void**, int, const int*);
    typedef int (*Destroy)(void*);
    typedef int (*Group)(void);
    typedef int (*SendRecv)(void*, size_t, int, int, void*, hipStream_t);      // ncclSend's buffer is const void*: same ABI
    typedef const char* (*ErrStr)(int);
    InitAll init_all = nullptr; Destroy destroy = nullptr; Group group_start = nullptr, group_end = nullptr;
    SendRecv send = nullptr, recv = nullptr; ErrStr err = nullptr;
    bool loaded = false;                // every symbol below resolved: set last, so a partial resolve can never be taken for a loaded library
    bool load(void** lib) {
        if (loaded) return true;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return false;
        Rccl r;
        r.init_all = (InitAll)dlsym(h, "ncclCommInitAll"); r.destroy = (Destroy)dlsym(h, "ncclCommDestroy");
        r.group_start = (Group)dlsym(h, "ncclGroupStart"); r.group_end = (Group)dlsym(h, "ncclGroupEnd");
        r.send = (SendRecv)dlsym(h, "ncclSend"); r.recv = (SendRecv)dlsym(h, "ncclRecv"); r.err = (ErrStr)dlsym(h, "ncclGetErrorString");
        if (!(r.init_all && r.destroy && r.group_start && r.group_end && r.send && r.recv)) { (void)dlclose(h); return false; }
        r.loaded = true;
        *this = r;
        *lib = h;
        return true;
    }
};
Rccl g_rccl;
constexpr int kNcclFloat = 7;       // ncclFloat32 (rccl.h ncclDataType_t)

}  // namespace

void crt_scene::drop_peers() {
    if (!rccl_comms.empty() && g_rccl.destroy)
        for (void* c : rccl_comms) if (c) (void)g_rccl.destroy(c);
    rccl_comms.clear();
    for (crt_scene* p : peers) delete p;
    peers.clear();
    if (!d_gather.empty() || !d_peer_tiles.empty() || !ev_peer.empty()) (void)hipSetDevice(device);
    for (float* b : d_gather) if (b) (void)hipFree(b);
    for (uint2* b : d_peer_tiles) if (b) (void)hipFree(b);
    for (hipEvent_t e : ev_peer) if (e) (void)hipEventDestroy(e);
    d_gather.clear(); d_peer_tiles.clear(); ev_peer.clear();
}

// a complete copy of `src` on another device: configuration by value, scene buffers over the fabric (hipMemcpyPeer), own stream,
// counters and events; frame buffers follow with its shard
static int replicate_scene(const crt_scene* src, int device, crt_scene** out) {
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    std::unique_ptr<crt_scene> owner(new (std::nothrow) crt_scene);
    crt_scene* r = owner.get();
    if (!r) return fail(CRT_ERR_NOMEM, "crt_set_devices: out of memory");
    r->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) r->n_cu = prop.multiProcessorCount;
    HIPCHK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    r->width = src->width; r->height = src->height; r->max_depth = src->max_depth; r->n_lights = src->n_lights;
    r->tex_width = src->tex_width; r->tex_height = src->tex_height; r->n_textures = src->n_textures;
    r->info = src->info; r->bvh2_stack = src->bvh2_stack; r->stack_entries = src->stack_entries; r->special_materials = src->special_materials;
    r->cam = src->cam; r->have_camera = src->have_camera; r->jitter = src->jitter;
    r->tri_min = src->tri_min; r->inplace_shadow = src->inplace_shadow; r->accel = src->accel; r->refill_min = src->refill_min; r->trace_pool = src->trace_pool; r->count_visits = src->count_visits;
    r->count_batched = src->count_batched;
    r->trace_occupancy = src->trace_occupancy; r->oversubscribe = src->oversubscribe; r->waves_per_workgroup = src->waves_per_workgroup;
    r->lanes_per_ray = src->lanes_per_ray; r->bounce_refill = src->bounce_refill; r->refill_pool = src->refill_pool; r->shadow_pool = src->shadow_pool; r->shadow_refill_min = src->shadow_refill_min; r->shadow_waves = src->shadow_waves; r->persistent = src->persistent; r->sort_shadow = src->sort_shadow;
    r->wave_samples = src->wave_samples; r->wide_first = src->wide_first; r->adaptive_tiles = src->adaptive_tiles; r->timing = src->timing;
    r->ray_bins = src->ray_bins; r->rows_padded = src->rows_padded;
    for (int k = 0; k < 3; ++k) { r->bounds_lo[k] = src->bounds_lo[k]; r->bounds_hi[k] = src->bounds_hi[k]; }
    r->scene_bufs = src->scene_bufs;
    r->shares_scene = device == src->device;
    if (device != src->device) {          // direct xGMI copies where the platform allows them; staged through the host otherwise
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(src->device, 0);
        (void)hipGetLastError();          // "already enabled" is not an error to report later
    }
    for (const auto& b : src->scene_bufs) {
        const void* from = *reinterpret_cast<void* const*>(reinterpret_cast<const char*>(src) + b.first);
        void** to = reinterpret_cast<void**>(reinterpret_cast<char*>(r) + b.first);
        if (!from) continue;
        if (device == src->device) { *to = const_cast<void*>(from); continue; }      // the same GPU again: one copy of the scene serves both
        hipError_t e = hipMalloc(to, b.second ? b.second : 16);
        if (e != hipSuccess) return fail(CRT_ERR_NOMEM, std::string("crt_set_devices: hipMalloc: ") + hipGetErrorString(e));
        if (b.second) HIPCHK(hipMemcpyPeer(*to, device, from, src->device, b.second));
    }
    int rc = finish_scene_setup(r);
    if (rc) return rc;
    *out = owner.release();
    return CRT_OK;
}

// base_rank / base_world: the shard of the frame the devices divide among themselves — the whole frame (0 of 1) for crt_set_devices,
// the caller's own shard (crt_set_shard) when option "streams" splits it over streams of one GPU: device k takes the tiles
// base_rank + k * base_world, + n * base_world, ... of the Morton order, i.e. every n-th tile of that shard's list starting at its k-th
int crt_set_devices(crt_scene* s, const int32_t* devices, uint32_t n_devices, uint32_t tile) {
    const int rc = set_devices_of_shard(s, devices, n_devices, tile, 0u, 1u);
    if (rc == CRT_OK) { s->shard_rank = 0u; s->shard_world = 1u; }      // the devices divide the whole frame (a refused call changes nothing)
    return rc;
}

static int set_devices_of_shard(crt_scene* s, const int32_t* devices, uint32_t n_devices, uint32_t tile, uint32_t base_rank, uint32_t base_world) {
    if (!s || !devices) return fail(CRT_ERR_INVALID, "crt_set_devices: null argument");
    if (s->primary) return fail(CRT_ERR_INVALID, "crt_set_devices: not on a replica");
    if (n_devices == 0 || n_devices > 64) return fail(CRT_ERR_INVALID, "crt_set_devices: 1..64 devices");
    if (tile < 8 || (tile & 7u) || tile > 1024) return fail(CRT_ERR_INVALID, "crt_set_devices: tile must be a multiple of 8 in 8..1024");
    if (devices[0] != s->device) return fail(CRT_ERR_INVALID, "crt_set_devices: devices[0] must be the device the scene was created on");
    int n_visible = 0;
    if (hipGetDeviceCount(&n_visible) != hipSuccess || n_visible <= 0) return fail(CRT_ERR_NO_DEVICE, "crt_set_devices: no HIP device visible");
    bool distinct = true;
    for (uint32_t k = 0; k < n_devices; ++k) {
        if (devices[k] < 0 || devices[k] >= n_visible) return fail(CRT_ERR_INVALID, "crt_set_devices: no such device");
        for (uint32_t j = 0; j < k; ++j) distinct = distinct && devices[j] != devices[k];
    }
    HIPCHK(hipSetDevice(s->device));
    if (n_devices > 1u && s->accel == 0u) { const int prc = ensure_planes(s); if (prc) return prc; }      // replicas copy (or share) the planes (a scene on its BVH2: ensure_planes, later)
    HIPCHK(hipStreamSynchronize(s->stream));
    s->drop_peers();
    s->streams = 1;
    int rc = CRT_OK;
    auto undo = [&](int code) { s->drop_peers(); (void)hipSetDevice(s->device); s->rank = base_rank; s->world = base_world; (void)alloc_frame_buffers(s); return code; };
    try {
        for (uint32_t k = 1; k < n_devices; ++k) {
            crt_scene* p = nullptr;
            if ((rc = replicate_scene(s, devices[k], &p))) return undo(rc);
            p->primary = s;
            s->peers.push_back(p);
            device_share(k, n_devices, base_rank, base_world, &p->rank, &p->world);
            p->tile = tile;
            if ((rc = alloc_frame_buffers(p))) return undo(rc);
        }
        if (hipSetDevice(s->device) != hipSuccess) return undo(fail(CRT_ERR_HIP, "crt_set_devices: hipSetDevice failed"));
        device_share(0u, n_devices, base_rank, base_world, &s->rank, &s->world);
        s->tile = tile;
        if ((rc = alloc_frame_buffers(s))) return undo(rc);
        for (crt_scene* p : s->peers) {                      // where a peer's slice lands on this device, and its tile list
            float* g = nullptr; uint2* t = nullptr; hipEvent_t e = nullptr;
            if ((rc = dev_alloc(&g, 3 * (size_t)p->n_local_pixels))) return undo(rc);
            s->d_gather.push_back(g);
            if ((rc = dev_alloc(&t, p->n_local_tiles))) return undo(rc);
            s->d_peer_tiles.push_back(t);
            if (p->n_local_tiles && hipMemcpy(t, p->tiles.data(), p->tiles.size() * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess)
                return undo(fail(CRT_ERR_HIP, "crt_set_devices: hipMemcpy H2D failed"));
            // recorded on the peer's stream, waited for on this device's: an event belongs to the device it was created on
            if (hipSetDevice(p->device) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
                return undo(fail(CRT_ERR_HIP, "crt_set_devices: hipEventCreate failed"));
            s->ev_peer.push_back(e);
            if (hipSetDevice(s->device) != hipSuccess) return undo(fail(CRT_ERR_HIP, "crt_set_devices: hipSetDevice failed"));
        }
    } catch (const std::exception& e) {
        return undo(fail(CRT_ERR_NOMEM, std::string("crt_set_devices: ") + e.what()));
    }
    // RCCL communicators, one per device, created by this one thread (ncclCommInitAll).  Virtual devices (the same GPU listed more
    // than once: a test arrangement) cannot have them; a missing library is not an error either: the copies below do the same job.
    s->gather_transport = 1;
    if (n_devices > 1 && distinct) {
        const char* env_tr = std::getenv("CRT_GATHER_TRANSPORT");     // "copy": peer copies even where RCCL would load
        if (env_tr && !std::strcmp(env_tr, "copy")) {
            (void)fail(CRT_OK, "crt_set_devices: CRT_GATHER_TRANSPORT=copy, gathering with hipMemcpyPeerAsync");
        } else if (!g_rccl.load(&s->rccl_lib)) {
            (void)fail(CRT_OK, "crt_set_devices: librccl.so not loadable, gathering with hipMemcpyPeerAsync");
        } else {
            std::vector<int> ids(devices, devices + n_devices);
            s->rccl_comms.assign(n_devices, nullptr);
            const int nr = g_rccl.init_all(s->rccl_comms.data(), (int)n_devices, ids.data());
            if (nr != 0) {
                s->rccl_comms.clear();
                (void)fail(CRT_OK, std::string("crt_set_devices: ncclCommInitAll failed (") + (g_rccl.err ? g_rccl.err(nr) : "?") + "), gathering with hipMemcpyPeerAsync");
            } else {
                s->gather_transport = 0;
            }
            (void)hipSetDevice(s->device);                  // ncclCommInitAll visits every device; the scene is complete either way
        }
    }
    return CRT_OK;
}

// [host] the tile list a scene would give logical device `device` of `n_devices` when those divide shard (rank of world) of a width x
// height frame among themselves — crt_set_shard (n_devices = 1), crt_set_devices (rank 0 of world 1) and option "streams" all deal
// tiles through this.  No GPU involved: the multi-GPU bookkeeping can be checked on any machine.
int crt_shard_tiles(uint32_t width, uint32_t height, uint32_t tile, uint32_t rank, uint32_t world, uint32_t device, uint32_t n_devices,
                    uint32_t* tile_xy, size_t capacity, size_t* n_tiles) {
    if (!n_tiles) return fail(CRT_ERR_INVALID, "crt_shard_tiles: null n_tiles");
    if (width == 0 || height == 0 || width > 65535u * 8u || height > 65535u * 8u) return fail(CRT_ERR_INVALID, "crt_shard_tiles: bad resolution");
    if (tile < 8 || (tile & 7u) || tile > 1024) return fail(CRT_ERR_INVALID, "crt_shard_tiles: tile must be a multiple of 8 in 8..1024");
    if (world == 0 || rank >= world || n_devices == 0 || n_devices > 64 || device >= n_devices) return fail(CRT_ERR_INVALID, "crt_shard_tiles: rank/world/device");
    uint32_t r = 0, w = 1;
    device_share(device, n_devices, rank, world, &r, &w);
    std::vector<uint2> tiles;
    try { deal_tiles(width, height, tile, r, w, tiles, nullptr); } catch (const std::exception& e) { return fail(CRT_ERR_NOMEM, std::string("crt_shard_tiles: ") + e.what()); }
    *n_tiles = tiles.size();
    if (tile_xy)
        for (size_t i = 0; i < tiles.size() && i < capacity; ++i) { tile_xy[2 * i] = tiles[i].x; tile_xy[2 * i + 1] = tiles[i].y; }
    return CRT_OK;
}

int crt_get_devices(crt_scene* s, uint32_t* n_devices, int32_t* devices, uint32_t capacity, int32_t* transport, float* last_gather_ms) {
    if (!s) return fail(CRT_ERR_INVALID, "crt_get_devices: null scene");
    if (n_devices) *n_devices = (uint32_t)s->peers.size() + 1u;
    if (devices)
        for (uint32_t k = 0; k < capacity && k <= s->peers.size(); ++k) devices[k] = k == 0 ? s->device : s->peers[k - 1]->device;
    if (transport) *transport = (int32_t)s->gather_transport;
    if (last_gather_ms) *last_gather_ms = s->last_gather_ms;
    return CRT_OK;
}

// every peer's packed sum buffer -> d_gather[k] on the primary device, complete in the primary's stream order when this returns
static int gather_peers(crt_scene* s) {
    const auto t0 = std::chrono::steady_clock::now();
    if (s->gather_transport == 0 && !s->rccl_comms.empty()) {
        // grouped point-to-point: device k sends on its own stream (after its frames), device 0 receives on its stream; each slice
        // crosses the one xGMI link between its GPU and this one
        int nr = g_rccl.group_start();
        for (size_t k = 0; k < s->peers.size() && nr == 0; ++k) {
            crt_scene* p = s->peers[k];
            const size_t count = 3 * (size_t)p->n_local_pixels;
            if (!count) continue;
            nr = g_rccl.send(p->d_sum, count, kNcclFloat, 0, s->rccl_comms[k + 1], p->stream);
            if (nr == 0) nr = g_rccl.recv(s->d_gather[k], count, kNcclFloat, (int)k + 1, s->rccl_comms[0], s->stream);
        }
        const int ne = g_rccl.group_end();
        if (nr == 0) nr = ne;
        (void)hipSetDevice(s->device);
        if (nr != 0) {
            // This read fails and the NEXT one gathers with peer copies.  Nothing is waited for here: a failed group can leave unmatched
            // send / recv kernels on the devices' streams, and a synchronise behind them may never return.  (The multi-device path has run
            // on virtual devices and at world size 1 only: DESIGN.md section 7.)
            s->gather_transport = 1;
            return fail(CRT_ERR_HIP, std::string("RCCL gather failed (") + (g_rccl.err ? g_rccl.err(nr) : "?") +
                                     "); this read-back is lost, the next one gathers with hipMemcpyPeerAsync (option gather_transport 1)");
        }
    }
    if (s->gather_transport != 0 || s->rccl_comms.empty()) {
        for (size_t k = 0; k < s->peers.size(); ++k) {
            crt_scene* p = s->peers[k];
            const size_t bytes = 3 * (size_t)p->n_local_pixels * sizeof(float);
            if (!bytes) continue;
            HIPCHK(hipSetDevice(p->device));
            if (p->device == s->device) HIPCHK(hipMemcpyAsync(s->d_gather[k], p->d_sum, bytes, hipMemcpyDeviceToDevice, p->stream));   // a virtual device
            else HIPCHK(hipMemcpyPeerAsync(s->d_gather[k], s->device, p->d_sum, p->device, bytes, p->stream));
            HIPCHK(hipEventRecord(s->ev_peer[k], p->stream));
        }
        HIPCHK(hipSetDevice(s->device));
        for (size_t k = 0; k < s->peers.size(); ++k)
            if (s->peers[k]->n_local_pixels) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_peer[k], 0));
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    s->last_gather_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return CRT_OK;
}

// Debug/test hook: copy out one of the frame's ray queues as left by the last crt_render_frame.
// which: 0/1 = path-ray queue written for an even/odd segment, 2 = shadow-ray queue of the last segment.
int crt_debug_launch_form(crt_scene* s, int32_t* form) {
    if (!s || !form) return fail(CRT_ERR_INVALID, "crt_debug_launch_form: null argument");
    *form = s->last_launch_form;
    return CRT_OK;
}

// Measurement aid: enabled lanes per node step of the counting kernels (option count_visits), closest-hit walks in hist[0..64], any-hit
// walks in hist[65..129], accumulated since the previous call (which zeroes it).  Process-wide (one device symbol); null stops it.
int crt_debug_step_hist(crt_scene* s, unsigned long long* hist) {
    static unsigned long long* d_hist = nullptr;
    if (!s) return fail(CRT_ERR_INVALID, "crt_debug_step_hist: null scene");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (!hist) { crt::set_step_hist(nullptr); return CRT_OK; }
    if (!d_hist) {
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&d_hist), 130 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(d_hist, 0, 130 * sizeof(unsigned long long)));
    }
    HIPCHK(hipMemcpy(hist, d_hist, 130 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(d_hist, 0, 130 * sizeof(unsigned long long)));
    crt::set_step_hist(d_hist, s->step_hist_mode);
    return CRT_OK;
}

int crt_debug_launch_info(crt_scene* s, int32_t info[4]) {
    if (!s || !info) return fail(CRT_ERR_INVALID, "crt_debug_launch_info: null argument");
    info[0] = s->last_launch_form; info[1] = s->last_launch_wide | (s->last_launch_one_pass << 1); info[2] = s->last_launch_samples; info[3] = (int32_t)s->peers.size() + 1;
    return CRT_OK;
}

int crt_debug_time_graph(crt_scene* s, uint32_t n_frames, const float* rxy, uint32_t reps, float* ms_stream, float* ms_graph) {
    if (!s || !rxy || !ms_stream || !ms_graph || n_frames == 0 || (n_frames & 1u) || reps == 0)
        return fail(CRT_ERR_INVALID, "crt_debug_time_graph: bad argument (n_frames must be even: the counter banks alternate)");
    if (!s->peers.empty() || s->primary)
        return fail(CRT_ERR_INVALID, "crt_debug_time_graph: one stream only (option streams 1, no crt_set_devices): the capture records this scene's stream, not its peers'");
    HIPCHK(hipSetDevice(s->device));
    const uint32_t timing0 = s->timing;
    s->timing = 0;                                         // event-carrying launches are not capturable
    int rc = crt_render_frame(s, rxy[0], rxy[1]);          // allocates, clears the counter banks once
    if (!rc) rc = crt_render_frame(s, rxy[0], rxy[1]);
    if (rc) { s->timing = timing0; return rc; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    auto done = [&](int code) {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        s->timing = timing0;
        return code;
    };
#define G_CHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return done(fail(CRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); } while (0)
    G_CHK(hipEventCreate(&e0));
    G_CHK(hipEventCreate(&e1));
    G_CHK(hipEventRecord(e0, s->stream));
    for (uint32_t r = 0; r < reps; ++r)
        for (uint32_t f = 0; f < n_frames; ++f)
            if ((rc = crt_render_frame_async(s, rxy[2 * f], rxy[2 * f + 1]))) return done(rc);
    G_CHK(hipEventRecord(e1, s->stream));
    G_CHK(hipStreamSynchronize(s->stream));
    float ms = 0.f;
    G_CHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_stream = ms / (float)(reps * n_frames);
    G_CHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    s->capturing = true;
    for (uint32_t f = 0; f < n_frames; ++f)
        if ((rc = crt_render_frame_async(s, rxy[2 * f], rxy[2 * f + 1]))) { s->capturing = false; (void)hipStreamEndCapture(s->stream, &graph); return done(rc); }
    s->capturing = false;
    G_CHK(hipStreamEndCapture(s->stream, &graph));
    G_CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    G_CHK(hipGraphLaunch(exec, s->stream));                // warm
    G_CHK(hipStreamSynchronize(s->stream));
    G_CHK(hipEventRecord(e0, s->stream));
    for (uint32_t r = 0; r < reps; ++r) G_CHK(hipGraphLaunch(exec, s->stream));
    G_CHK(hipEventRecord(e1, s->stream));
    G_CHK(hipStreamSynchronize(s->stream));
    G_CHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_graph = ms / (float)(reps * n_frames);
#undef G_CHK
    return done(CRT_OK);
}

int crt_debug_read_queue(crt_scene* s, int which, uint32_t segment, crt_ray* dst, size_t cap, size_t* n_out) {
    if (!s || !n_out) return fail(CRT_ERR_INVALID, "crt_debug_read_queue: null argument");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (!s->frame_buffers_ready || segment > 16) return fail(CRT_ERR_INVALID, "crt_debug_read_queue: no frame rendered");
    std::vector<uint32_t> counts(kCounters);
    HIPCHK(hipMemcpy(counts.data(), s->counts(), kCounters * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (which != 2 && s->d_bins && s->ray_bins) return fail(CRT_ERR_INVALID, "crt_debug_read_queue: the path-ray queue is binned (option ray_bins 0 gives the sub-queue form this reads)");
    const uint32_t first_def = first_deferred_segment(s);
    const float4* src = which == 2 ? ((s->d_nee && segment >= first_def && segment - first_def < s->defer_regions) ? s->d_nee + 2 * (size_t)(segment - first_def) * 8u * s->sub_capacity : nullptr)
                                   : s->d_rays[segment & 1];
    if (!src) return fail(CRT_ERR_INVALID, "crt_debug_read_queue: that queue does not exist (shadow rays: only of segments that defer them, inplace_shadow 0 / 2; path queues: max_depth > 1)");
    const size_t entry = sizeof(crt_ray);        // a deferred shadow ray is (o, tmax) (d, contribution slot)
    size_t total = 0;
    for (uint32_t g = 0; g < 8; ++g) total += counts[counter_index(segment, which == 2 ? 1 : 0, g)];
    *n_out = total;
    if (dst) {
        if (!src) return fail(CRT_ERR_INVALID, "crt_debug_read_queue: that queue does not exist at max_depth 1");
        size_t done = 0;
        for (uint32_t g = 0; g < 8 && done < cap; ++g) {
            size_t n = counts[counter_index(segment, which == 2 ? 1 : 0, g)];
            n = std::min(n, cap - done);
            if (!n) continue;
            const char* from = reinterpret_cast<const char*>(src) + (size_t)g * s->sub_capacity * entry;
            HIPCHK(hipMemcpy2D(dst + done, sizeof(crt_ray), from, entry, sizeof(crt_ray), n, hipMemcpyDeviceToHost));
            done += n;
        }
    }
    return CRT_OK;
}

int crt_trace_device(crt_scene* s, const void* d_rays, size_t n, void* d_hits, int mode, void* d_stats, int sync) {
    if (!s || !d_rays || !d_hits) return fail(CRT_ERR_INVALID, "crt_trace_device: null argument");
    if (mode < 0 || mode > (CRT_TRACE_ANY | CRT_TRACE_BVH2 | CRT_TRACE_TIE_LOWEST_ID)) return fail(CRT_ERR_INVALID, "crt_trace_device: bad mode");
    const bool any_hit = (mode & CRT_TRACE_ANY) != 0;
    if (n >= (1ull << 31)) return fail(CRT_ERR_LIMIT, "crt_trace_device: too many rays for one launch");
    HIPCHK(hipSetDevice(s->device));
    if (n == 0) return CRT_OK;
    crt::TraceArgs ta{};
    ta.nodes = s->d_nodes; ta.tris = s->d_tris;
    ta.rays = static_cast<const float4*>(d_rays); ta.hits = static_cast<float4*>(d_hits);
    ta.stats = static_cast<uint32_t*>(d_stats); ta.count_ptr = nullptr; ta.n = (uint32_t)n; ta.out_orig_id = 1;
    ta.stack_entries = s->stack_entries;
    ta.refill_min = s->refill_min;
    ta.tri_min = s->tri_min;
    ta.overflow = s->d_overflow;
    ta.pool_split_log2 = s->trace_pool == 64u ? 2u : s->trace_pool == 128u ? 1u : 0u;
    s->n_spans = 0;
    if ((mode & CRT_TRACE_BVH2) && !s->d_bvh2) return fail(CRT_ERR_INVALID, "crt_trace: the scene was created without a BVH2 (desc.bvh)");
    EventSpan* sp = s->new_span(any_hit ? 2 : 1);
    if (sp) crt::set_launch_events(sp->a, sp->b);
    if (mode & CRT_TRACE_BVH2) {
        crt::Bvh2Args ba{};
        ba.nodes = s->d_bvh2; ba.tris = s->d_tris2; ba.rays = ta.rays; ba.hits = ta.hits; ba.stats = ta.stats;
        ba.n = (uint32_t)n; ba.tie = (mode & CRT_TRACE_TIE_LOWEST_ID) ? 1u : 0u; ba.stack_entries = s->bvh2_stack; ba.overflow = s->d_overflow;
        const uint32_t g = s->trace_grid(n, 8);          // every chunk has its workgroup (CRT_CHUNK_LOOP)
        crt::launch_trace_bvh2(ba, any_hit, d_stats != nullptr, g, s->waves_per_workgroup, s->stream);
    } else {
        crt::launch_trace(ta, any_hit ? 1 : 0, d_stats != nullptr, s->trace_grid(n, 8, 1024), s->waves_per_workgroup, s->stream);
    }
    HIPCHK(hipGetLastError());
    s->stats_pending = true;
    s->stats_from_frame = false;
    s->stats.closest_rays = any_hit ? 0 : n;
    s->stats.any_rays = any_hit ? n : 0;
    if (sync) HIPCHK(hipStreamSynchronize(s->stream));
    return CRT_OK;
}

int crt_trace(crt_scene* s, const crt_ray* rays, size_t n, crt_hit* hits, int mode, crt_ray_stats* stats) {
    if (!s || (n && (!rays || !hits))) return fail(CRT_ERR_INVALID, "crt_trace: null argument");
    HIPCHK(hipSetDevice(s->device));
    if (n == 0) return CRT_OK;
    if (n > s->t_cap) {
        if (s->d_t_rays) hipFree(s->d_t_rays);
        if (s->d_t_hits) hipFree(s->d_t_hits);
        if (s->d_t_stats) hipFree(s->d_t_stats);
        s->d_t_rays = nullptr; s->d_t_hits = nullptr; s->d_t_stats = nullptr; s->t_cap = 0;
        int rc;
        if ((rc = dev_alloc(&s->d_t_rays, 2 * n))) return rc;
        if ((rc = dev_alloc(&s->d_t_hits, n))) return rc;
        if ((rc = dev_alloc(&s->d_t_stats, n))) return rc;
        s->t_cap = n;
    }
    static_assert(sizeof(crt_ray) == 32 && sizeof(crt_hit) == 16 && sizeof(crt_ray_stats) == 4, "ray/hit layout");
    HIPCHK(hipMemcpyAsync(s->d_t_rays, rays, n * sizeof(crt_ray), hipMemcpyHostToDevice, s->stream));
    int rc = crt_trace_device(s, s->d_t_rays, n, s->d_t_hits, mode, stats ? s->d_t_stats : nullptr, 0);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(hits, s->d_t_hits, n * sizeof(crt_hit), hipMemcpyDeviceToHost, s->stream));
    if (stats) HIPCHK(hipMemcpyAsync(stats, s->d_t_stats, n * sizeof(crt_ray_stats), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CRT_OK;
}

}  // extern "C"
