// Regrouping rays between launches (BASELINE configs[3], "divergence / sorting stress"): part of rt_kernels.hip (included there, inside
// namespace crt, behind wave_append and the scheduling helpers).  Both forms are built, bit-identical and OFF by default — each was measured
// to cost more than the fuller waves return (profiles/r04_experiments.md section 4, profiles/r05_experiments.md section 1d):
//   ray_bins     the rays a segment emits are appended to 4096 bins keyed by (direction octant, 8^3 cell of the origin)
//   sort_shadow  a counting sort of the frame's deferred NEE shadow rays by the 16^3 Morton cell of their origin
#pragma once

// ---- bounce-ray bins (rt_kernels.hpp RayBins) ----
__device__ __forceinline__ uint32_t ray_bin_key(const RayBins& b, float4 o, float4 d) {
    // NaN coordinates fall into cell 0 (v_max ignores a NaN operand); the key only decides where the ray waits, never what it hits
    const float cx = __builtin_fminf(__builtin_fmaxf((o.x - b.origin[0]) * b.scale[0], 0.0f), 7.0f);
    const float cy = __builtin_fminf(__builtin_fmaxf((o.y - b.origin[1]) * b.scale[1], 0.0f), 7.0f);
    const float cz = __builtin_fminf(__builtin_fmaxf((o.z - b.origin[2]) * b.scale[2], 0.0f), 7.0f);
    const uint32_t oct = (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u);
    return (oct << 9) | ((uint32_t)cz << 6) | ((uint32_t)cy << 3) | (uint32_t)cx;      // octant-major: neighbouring bins share the octant
}
// Queue entry for this lane's ray (meaningless where !want).  The lanes of a wave that share a key are found with one ballot per
// distinct key (a bounce off one 8 x 8 pixel patch spreads over a handful of cells and 4 - 8 octants), each such set takes its places
// with ONE atomic (all sets' atomics issue together), and what does not fit its bin takes a place in the overflow region.
// `tab`: 768 words of wave-private LDS that nothing else uses right now (the wave's traversal stack: both walks are over when a
// segment emits its rays), or null.
__device__ __forceinline__ uint32_t bin_append(const RayBins& b, bool want, uint32_t key, uint32_t* tab = nullptr) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = 0;
    if (b.per_lane == 2u && tab != nullptr) {
        // Ranking through LDS: a 256-slot wave-private table hashed by the key.  The first ray to reach a slot owns
        // it and publishes its key; the rays that share that key then rank themselves with one ds_add_rtn each and take their places in
        // the bin with ONE global atomic for all of them.  A ray whose key collides with another key's slot (a wave of bounce rays holds
        // ~50 keys in 256 slots: a few rays per wave) takes its place with an atomic of its own, as per_lane = 1 does for every ray.
        // ~45 instructions per emitting wave where the ballot loop below needs ~12 per distinct key (~600 on a wave of bounce rays).
        uint32_t* const t_cnt = tab; uint32_t* const t_key = tab + 256; uint32_t* const t_base = tab + 512;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) t_cnt[lane + 64u * k] = 0u;
        __builtin_amdgcn_wave_barrier();
        const uint32_t h = (key ^ (key >> 8)) & 255u;               // cell bits and octant bits folded together
        const bool first = want && atomicAdd(&t_cnt[h], 1u) == 0u;   // ds_add_rtn_u32: who got here first owns the slot
        if (first) t_key[h] = key;
        __builtin_amdgcn_wave_barrier();
        const bool same = want && t_key[h] == key;                    // this ray shares the owner's key
        __builtin_amdgcn_wave_barrier();
        if (first) t_cnt[h] = 0u;
        __builtin_amdgcn_wave_barrier();
        uint32_t rank = 0;
        if (same) rank = atomicAdd(&t_cnt[h], 1u);                    // consecutive ranks among the rays of that key
        __builtin_amdgcn_wave_barrier();
        if (same && rank == 0u) t_base[h] = atomicAdd(b.count + key, t_cnt[h]);
        __builtin_amdgcn_wave_barrier();
        if (same) idx = t_base[h] + rank;
        else if (want) idx = atomicAdd(b.count + key, 1u);            // a collided key: its own place
        __builtin_amdgcn_wave_barrier();                              // the table is the traversal stack again after this
    } else if (b.per_lane) {
        // a wave of bounce rays holds about as many keys as rays (its rays left one cell in one octant and landed all over the
        // scene): finding the few lanes that share one costs more than their atomics
        if (want) idx = atomicAdd(b.count + key, 1u);
    } else {
        const unsigned long long below = (1ull << lane) - 1ull;
        unsigned long long todo = __ballot(want);
        uint32_t rank = 0, n_same = 0, leader = lane;
        while (todo) {                                               // wave-uniform: one pass per distinct key
            const int l = __builtin_ctzll(todo);
            const uint32_t k = __builtin_amdgcn_readlane(key, l);
            const unsigned long long m = __ballot(want && key == k);
            if (want && key == k) { rank = (uint32_t)__builtin_popcountll(m & below); n_same = (uint32_t)__builtin_popcountll(m); leader = (uint32_t)l; }
            todo &= ~m;
        }
        uint32_t base = 0;
        if (want && lane == leader) base = atomicAdd(b.count + key, n_same);
        base = __shfl(base, (int)leader);
        idx = base + rank;
    }
    const uint32_t cap = want ? b.cap[key] : 0u;
    const bool fits = want && idx < cap;
    const uint32_t ov = wave_append(want && !fits, b.ovf_count);
    return fits ? b.off[key] + idx : b.ovf_base + ov;
}
// consumer index -> queue entry: the bin whose range [start[b], start[b + 1]) holds e (binary search, 12 steps), or the overflow region
__device__ __forceinline__ uint32_t binned_entry(const uint32_t* __restrict__ start, const uint32_t* __restrict__ off, uint32_t ovf_base, uint32_t e) {
    const uint32_t in_bins = start[CRT_RAY_BINS];
    if (e >= in_bins) return ovf_base + (e - in_bins);
    uint32_t lo = 0u, hi = CRT_RAY_BINS;                             // invariant: start[lo] <= e < start[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (start[mid] <= e) lo = mid; else hi = mid;
    }
    return off[lo] + (e - start[lo]);
}

// Between the launch that fills the bins of a segment and the launch that walks them (RayBins in rt_kernels.hpp): fill counts ->
// consumer index space, ray count of the consuming launch, next frame's capacities and offsets, counters back to zero.
__global__ void __launch_bounds__(1024) k_bin_scan(BinScanArgs a) {
    __shared__ uint32_t s_wave[2][16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t PER = CRT_RAY_BINS / 1024u;
    uint32_t filled[PER], want[PER], f_sum = 0, w_sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t b = tid * PER + k, c = a.count[b], cap = a.cap[b];
        filled[k] = c < cap ? c : cap;
        want[k] = c + (c >> 3) + 16u;                    // what the bin received, an eighth more, and room for a bin that was empty
        f_sum += filled[k]; w_sum += want[k];
        a.count[b] = 0u;
    }
    // exclusive prefix of (f_sum, w_sum) over the 1024 threads: inside the wave by shuffles, across waves through LDS
    uint32_t f_inc = f_sum, w_inc = w_sum;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t fu = __shfl_up(f_inc, d), wu = __shfl_up(w_inc, d);
        if ((int)lane >= d) { f_inc += fu; w_inc += wu; }
    }
    if (lane == 63u) { s_wave[0][wave] = f_inc; s_wave[1][wave] = w_inc; }
    __syncthreads();
    uint32_t f_base = 0, w_base = 0, f_all = 0, w_all = 0;
    for (uint32_t w = 0; w < 16u; ++w) {
        if (w < wave) { f_base += s_wave[0][w]; w_base += s_wave[1][w]; }
        f_all += s_wave[0][w]; w_all += s_wave[1][w];
    }
    uint32_t f_at = f_base + f_inc - f_sum, w_at = w_base + w_inc - w_sum;
    // the capacities asked for may exceed the bins' share of the queue (a launch of more samples than the last one): scaled down
    const float shrink = w_all > a.queue_entries ? (float)a.queue_entries / (float)w_all : 1.0f;
    if (shrink < 1.0f) {
        // rescale and redo the offsets' prefix on the scaled values
        uint32_t s_sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) { want[k] = (uint32_t)((float)want[k] * shrink); s_sum += want[k]; }
        uint32_t s_inc = s_sum;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(s_inc, d); if ((int)lane >= d) s_inc += u; }
        __syncthreads();
        if (lane == 63u) s_wave[1][wave] = s_inc;
        __syncthreads();
        w_base = 0;
        for (uint32_t w = 0; w < wave; ++w) w_base += s_wave[1][w];
        w_at = w_base + s_inc - s_sum;
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t b = tid * PER + k;
        a.start[b] = f_at; f_at += filled[k];
        a.cap_next[b] = want[k]; a.off_next[b] = w_at; w_at += want[k];
    }
    if (tid == 0u) {
        a.start[CRT_RAY_BINS] = f_all;
        *a.n_in = f_all + *a.ovf_count;
        *a.ovf_count = 0u;
    }
}

// ---- the deferred shadow rays SORTED by where they start (option "sort_shadow"; BASELINE configs[3], "sorting stress") ----
// The NEE rays of a frame all aim at the lights; what differs is where they start — and a wave of the queue's emission order holds rays that
// start all over the scene (the hit points of one wave's bounce rays).  A counting sort by the 16 x 16 x 16 cell of the origin (Morton
// order of the cells) puts rays that start together AND head the same way into one wave: k_nee_hist (per-block LDS histogram -> global),
// k_nee_scan (4096 bins -> starts, and the sorted array cut into 8 equal parts, one per XCD group), k_nee_scatter (a block reserves its
// places in every bin with one atomic per bin, ranks its rays through LDS, writes perm[place] = queue entry).  k_shadow_deferred then draws
// entry perm[i] for i in sorted order.  Which lane walks a ray changes; the ray, its walk and its slot do not: the same bits.
__device__ __forceinline__ uint32_t nee_cell_key(const NeeSortArgs& a, float4 o) {
    // NaN coordinates fall into cell 0 (v_max ignores a NaN operand); the key only decides where the ray waits, never what it hits
    const uint32_t cx = (uint32_t)__builtin_fminf(__builtin_fmaxf((o.x - a.origin[0]) * a.scale[0], 0.0f), 15.0f);
    const uint32_t cy = (uint32_t)__builtin_fminf(__builtin_fmaxf((o.y - a.origin[1]) * a.scale[1], 0.0f), 15.0f);
    const uint32_t cz = (uint32_t)__builtin_fminf(__builtin_fmaxf((o.z - a.origin[2]) * a.scale[2], 0.0f), 15.0f);
    auto spread = [](uint32_t v) { v = (v | (v << 4)) & 0x0c3u; v = (v | (v << 2)) & 0x249u; return v; };      // 4 bits -> every third bit
    return spread(cx) | (spread(cy) << 1) | (spread(cz) << 2);
}
#define CRT_NEE_BINS 4096u
// every block walks the same slice of every sub-queue in both passes: entries blockIdx * 256 + tid, + gridDim * 256, ...
template <typename F>
__device__ __forceinline__ void nee_for_each(const NeeSortArgs& a, F f) {
    for (uint32_t q = 0; q < a.n_queues; ++q) {
        const uint32_t n = a.count[(size_t)(q >> 3) * a.count_stride + (q & 7u) * CRT_COUNTER_STRIDE];
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) f(q * a.sub_capacity + i);
    }
}
__global__ void __launch_bounds__(256) k_nee_hist(NeeSortArgs a) {
    __shared__ uint32_t s_hist[CRT_NEE_BINS];
    for (uint32_t b = threadIdx.x; b < CRT_NEE_BINS; b += blockDim.x) s_hist[b] = 0u;
    __syncthreads();
    nee_for_each(a, [&](uint32_t e) { atomicAdd(&s_hist[nee_cell_key(a, a.shadow[2 * (size_t)e])], 1u); });
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < CRT_NEE_BINS; b += blockDim.x)
        if (s_hist[b]) atomicAdd(a.hist + b, s_hist[b]);
}
// one block of 1024 threads: exclusive scan of the bins -> cursor[bin] = first place of the bin; hist back to zero for the next frame;
// meta[k * CRT_COUNTER_STRIDE] (k = 0..7) = rays in the k-th eighth of the sorted array, meta[8 * CRT_COUNTER_STRIDE] = length of an eighth
__global__ void __launch_bounds__(1024) k_nee_scan(NeeSortArgs a) {
    __shared__ uint32_t s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t c[4], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) { c[k] = a.hist[tid * 4u + k]; a.hist[tid * 4u + k] = 0u; sum += c[k]; }
    uint32_t inc = sum;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d); if ((int)lane >= d) inc += u; }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (uint32_t w = 0; w < 16u; ++w) { if (w < wave) base += s_wave[w]; total += s_wave[w]; }
    uint32_t at = base + inc - sum;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) { a.cursor[tid * 4u + k] = at; at += c[k]; }
    if (tid < 9u) {
        const uint32_t eighth = (((total + 7u) >> 3) + 255u) & ~255u;
        if (tid == 8u) a.meta[8u * CRT_COUNTER_STRIDE] = eighth;
        else {
            const uint32_t lo = tid * eighth;
            a.meta[tid * CRT_COUNTER_STRIDE] = lo >= total ? 0u : (total - lo < eighth ? total - lo : eighth);
        }
    }
}
__global__ void __launch_bounds__(256) k_nee_scatter(NeeSortArgs a) {
    __shared__ uint32_t s_base[CRT_NEE_BINS];      // pass 1: this block's rays per bin; then: the block's first place in the bin
    __shared__ uint32_t s_rank[CRT_NEE_BINS];
    for (uint32_t b = threadIdx.x; b < CRT_NEE_BINS; b += blockDim.x) { s_base[b] = 0u; s_rank[b] = 0u; }
    __syncthreads();
    nee_for_each(a, [&](uint32_t e) { atomicAdd(&s_base[nee_cell_key(a, a.shadow[2 * (size_t)e])], 1u); });
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < CRT_NEE_BINS; b += blockDim.x)
        if (s_base[b]) s_base[b] = atomicAdd(a.cursor + b, s_base[b]);
    __syncthreads();
    nee_for_each(a, [&](uint32_t e) {
        const uint32_t key = nee_cell_key(a, a.shadow[2 * (size_t)e]);
        a.perm[s_base[key] + atomicAdd(&s_rank[key], 1u)] = e;
    });
}

