// Split-BVH construction.  Behavioural contract: Caitlyn/sbvh.h (cited per function).
// The output (node boxes, BFS numbering, leaf order, duplicate references) has to be
// the tree the reference would have built, because everything downstream — CWBVH
// conversion, triangle slots, hit IDs — is keyed on it.  tests/test_host.py pins it against
// the known-answer values of SURVEY.md §8c (Cornell tree entry for entry; the n = 8 and n = 183 spatial-split probes).
#include "sbvh.hpp"

#include <sched.h>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <future>
#include <thread>

namespace crt {
namespace {

constexpr int kBins = 256;            // sbvh.h:17
constexpr int kLeafRefs = 2;          // sbvh.h:93
constexpr float kSplitAlpha = 0.00001f;   // sbvh.h:96
constexpr float kHuge = 1e20f;        // sbvh.h:13 (`inf`), stored into floats

struct Ref { int id; Aabb box; };                 // FlatNode.h:7-11
struct Spec { int n = 0; Aabb box; };             // FlatNode.h:13-17

struct BuildNode {
    Aabb box;
    int child[2] = {-1, -1};
    uint32_t start = 0, range = 0;
    bool leaf = false;
};

struct ObjectSplit { float sah = kHuge; int dim = 0; int n_left = 0; Aabb lb, rb; };   // sbvh.h:21-28
struct SpatialSplit { float sah = kHuge; int dim = 0; float pos = 0.f; };              // sbvh.h:30-35
struct Bin { Aabb box; int enter = 0, exit = 0; };                                     // sbvh.h:37-42

// float -> int the way the reference's x86 build does it (cvttss2si): out-of-range and
// NaN give INT_MIN.  Needed because flat nodes produce 0*inf = NaN bin coordinates
// (sbvh.h:428, :445-446) and the clamp that follows turns INT_MIN into bin 0.
inline int trunc_i32(float f) {
    if (!(f > -2147483648.0f && f < 2147483648.0f)) return INT_MIN;
    return (int)f;
}
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }   // sbvh.h:44-47
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : v > hi ? hi : v; }

struct Builder {
    const crt_triangle* tris;
    const float3* verts;
    bool spatial_enabled;

    std::vector<Ref> refs;            // used as a stack: the node being split owns the tail
    std::vector<int32_t> leaf_ids;    // triangle_indices
    std::vector<Aabb> right_bounds;
    std::vector<BuildNode> nodes;
    std::vector<Bin> bins[3];
    float min_overlap = 0.f;

    // sbvh.h:326-336: order by centroid on `dim`, then by id (a strict total order
    // inside one node, so any sorting algorithm gives the same permutation).
    void sort_tail(int n, int dim) {
        std::sort(refs.end() - n, refs.end(), [dim](const Ref& a, const Ref& b) {
            float ca = a.box.centre()[dim], cb = b.box.centre()[dim];
            return (ca < cb) || (ca == cb && a.id < b.id);
        });
    }

    // sbvh.h:338-378: exact SAH sweep over all three axes.
    ObjectSplit find_object_split(const Spec& spec, float node_sah) {
        ObjectSplit best;
        const int n = spec.n;
        const size_t first = refs.size() - n;
        for (int dim = 0; dim < 3; ++dim) {
            sort_tail(n, dim);
            Aabb rb;
            for (int i = n - 1; i > 0; --i) {
                rb.grow(refs[first + i].box);
                right_bounds[i - 1] = rb;
            }
            Aabb lb;
            for (int i = 1; i < n; ++i) {
                lb.grow(refs[first + i - 1].box);
                float sah = node_sah + (lb.half_area() * (float)i + right_bounds[i - 1].half_area() * (float)(n - i));
                if (sah < best.sah) {
                    best.sah = sah;
                    best.dim = dim;
                    best.n_left = i;
                    best.lb = lb;
                    best.rb = right_bounds[i - 1];
                }
            }
        }
        return best;
    }

    // sbvh.h:391-422: clip one reference against the plane x[dim] = pos.
    void split_ref(Ref& l, Ref& r, const Ref& ref, int dim, float pos) const {
        l.id = r.id = ref.id;
        l.box = r.box = Aabb();
        const int32_t* vi = tris[ref.id].v;
        for (int i = 0; i < 3; ++i) {
            const float3 a = verts[vi[i]];
            const float3 b = verts[vi[(i + 1) % 3]];
            const float ap = a[dim], bp = b[dim];
            if (ap <= pos) l.box.grow(a);
            if (ap >= pos) r.box.grow(a);
            if ((ap < pos && bp > pos) || (ap > pos && bp < pos)) {
                float t = clampf((pos - ap) / (bp - ap), 0.0f, 1.0f);
                float3 x = a + t * (b - a);             // sbvh.h:63-68 lerp
                l.box.grow(x);
                r.box.grow(x);
            }
        }
        l.box.hi[dim] = pos;
        r.box.lo[dim] = pos;
        l.box.clip(ref.box);
        r.box.clip(ref.box);
    }

    // sbvh.h:424-495: chopped binning, 256 bins per axis.
    SpatialSplit find_spatial_split(const Spec& spec, float node_sah) {
        const float3 origin = spec.box.lo;
        const float3 bin_size = (spec.box.hi - origin) * (1.0f / (float)kBins);
        const float3 inv_bin(1.0f / bin_size.x, 1.0f / bin_size.y, 1.0f / bin_size.z);
        for (int dim = 0; dim < 3; ++dim)
            for (int i = 0; i < kBins; ++i) bins[dim][i] = Bin();

        for (size_t r = refs.size() - spec.n; r < refs.size(); ++r) {
            const Ref ref = refs[r];
            const float3 flo = (ref.box.lo - origin) * inv_bin;
            const float3 fhi = (ref.box.hi - origin) * inv_bin;
            int first[3], last[3];
            for (int d = 0; d < 3; ++d) {
                first[d] = clampi(trunc_i32(flo[d]), 0, kBins - 1);
                last[d] = clampi(trunc_i32(fhi[d]), first[d], kBins - 1);
            }
            for (int dim = 0; dim < 3; ++dim) {
                Ref cur = ref;
                for (int i = first[dim]; i < last[dim]; ++i) {
                    Ref l, rr;
                    split_ref(l, rr, cur, dim, origin[dim] + bin_size[dim] * (float)(i + 1));
                    bins[dim][i].box.grow(l.box);
                    cur = rr;
                }
                bins[dim][last[dim]].box.grow(cur.box);
                bins[dim][first[dim]].enter++;
                bins[dim][last[dim]].exit++;
            }
        }

        SpatialSplit best;
        for (int dim = 0; dim < 3; ++dim) {
            Aabb rb;
            for (int i = kBins - 1; i > 0; --i) {
                rb.grow(bins[dim][i].box);
                right_bounds[i - 1] = rb;
            }
            Aabb lb;
            int n_left = 0, n_right = spec.n;
            for (int i = 1; i < kBins; ++i) {
                lb.grow(bins[dim][i - 1].box);
                n_left += bins[dim][i - 1].enter;
                n_right -= bins[dim][i - 1].exit;
                // association as written at sbvh.h:484 (left to right)
                float sah = node_sah + lb.half_area() * (float)n_left + right_bounds[i - 1].half_area() * (float)n_right;
                if (sah < best.sah) {
                    best.sah = sah;
                    best.dim = dim;
                    best.pos = origin[dim] + bin_size[dim] * (float)i;
                }
            }
        }
        return best;
    }

    // Carry out a spatial split (behaviour of sbvh.h:497-569).  The node's references are the tail of `refs`; afterwards that tail
    // reads [left child's refs ..., right child's refs ...].  References that lie wholly on one side of the plane go there.  Each
    // straddler then meets one of three fates, whichever the SAH prices lowest against the children's boxes as they stand when
    // its turn comes: stay whole in the left child, stay whole in the right child, or be clipped at the plane into a reference for
    // each (the "split" of a split BVH: the triangle then sits in two leaves).  Ties go in that order.  The straddlers are taken in
    // the order the first sweep leaves them, which the outcome depends on, so both sweeps move elements exactly as the reference's do.
    void do_spatial_split(Spec& left, Spec& right, const Spec& spec, const SpatialSplit& plane) {
        const int axis = plane.dim;
        const float at = plane.pos;
        const size_t begin = refs.size() - spec.n;
        size_t lo = begin;              // refs[begin, lo): decided for the left child
        size_t hi = refs.size();        // refs[hi, end):   decided for the right child; refs[lo, hi) undecided
        left.box = right.box = Aabb();

        // sweep 1: everything that does not straddle the plane
        for (size_t i = lo; i < hi;) {
            const Aabb box = refs[i].box;
            if (box.hi[axis] <= at) {
                left.box.grow(box);
                std::swap(refs[i], refs[lo]);
                ++lo;
                ++i;
            } else if (box.lo[axis] >= at) {
                right.box.grow(box);
                --hi;
                std::swap(refs[i], refs[hi]);      // position i now holds an unexamined reference: look at it next
            } else {
                ++i;
            }
        }

        // sweep 2: the straddlers, refs[lo, hi), front to back
        auto priced = [](const Aabb& box, float count) { return box.half_area() * count; };
        while (lo < hi) {
            const Ref whole = refs[lo];
            Ref left_part, right_part;
            split_ref(left_part, right_part, whole, axis, at);
            const float n_left = (float)(lo - begin), n_right = (float)(refs.size() - hi);

            Aabb left_if_whole = left.box, right_if_whole = right.box, left_if_clipped = left.box, right_if_clipped = right.box;
            left_if_whole.grow(whole.box);
            right_if_whole.grow(whole.box);
            left_if_clipped.grow(left_part.box);
            right_if_clipped.grow(right_part.box);

            const float cost_whole_left = priced(left_if_whole, n_left + 1) + priced(right.box, n_right);
            const float cost_whole_right = priced(left.box, n_left) + priced(right_if_whole, n_right + 1);
            const float cost_clipped = priced(left_if_clipped, n_left + 1) + priced(right_if_clipped, n_right + 1);
            const float cheapest = fmin_(cost_whole_left, fmin_(cost_whole_right, cost_clipped));   // sbvh.h:58-61

            if (cheapest == cost_whole_left) {
                left.box = left_if_whole;
                ++lo;
            } else if (cheapest == cost_whole_right) {
                right.box = right_if_whole;
                --hi;
                std::swap(refs[lo], refs[hi]);
            } else {
                left.box = left_if_clipped;
                right.box = right_if_clipped;
                refs[lo] = left_part;
                ++lo;
                refs.push_back(right_part);         // the duplicate joins the right child's end of the tail
            }
        }
        left.n = (int)(lo - begin);
        right.n = (int)(refs.size() - hi);
    }

    void make_leaf(int node, const Spec& spec) {   // sbvh.h:190-205
        for (int i = 0; i < spec.n; ++i) {
            leaf_ids.push_back(refs.back().id);
            refs.pop_back();
        }
        BuildNode& n = nodes[node];
        n.box = spec.box;
        n.start = (uint32_t)(leaf_ids.size() - spec.n);
        n.range = (uint32_t)spec.n;
        n.leaf = true;
    }

    // One split decision on the tail of `refs` (the body of the loop at sbvh.h:250-279).  Afterwards the
    // tail is [left refs ..., right refs ...], duplicates of a spatial split included.
    void split_once(const Spec& spec, Spec& left, Spec& right) {
        const float node_area = spec.box.half_area();
        const float node_sah = 2.0f * node_area;
        ObjectSplit object = find_object_split(spec, node_sah);
        SpatialSplit spatial;
        if (spatial_enabled) {
            Aabb overlap = object.lb;
            overlap.clip(object.rb);
            if (overlap.half_area() >= min_overlap) spatial = find_spatial_split(spec, node_sah);
        }
        const float min_sah = fmin_(object.sah, spatial.sah);
        if (spatial_enabled && min_sah == spatial.sah) {
            do_spatial_split(left, right, spec, spatial);
        } else {                                   // sbvh.h:379-389
            sort_tail(spec.n, object.dim);
            left.n = object.n_left;
            left.box = object.lb;
            right.n = spec.n - object.n_left;
            right.box = object.rb;
        }
    }

    // sbvh.h:218-283.  The right child is pushed last and therefore built first: it owns
    // the tail of `refs`, which is what make_leaf pops.
    void run(const Spec& root) {
        struct Item { int node; Spec spec; };
        std::vector<Item> stack;
        nodes.emplace_back();
        stack.push_back({0, root});
        while (!stack.empty()) {
            Item top = stack.back();
            stack.pop_back();
            nodes[top.node].box = top.spec.box;
            nodes[top.node].leaf = false;
            if (top.spec.n <= kLeafRefs) {
                make_leaf(top.node, top.spec);
                continue;
            }
            Spec left, right;
            split_once(top.spec, left, right);
            const int l = (int)nodes.size();
            nodes.emplace_back();
            nodes.emplace_back();
            nodes[top.node].child[0] = l;
            nodes[top.node].child[1] = l + 1;
            stack.push_back({l, left});
            stack.push_back({l + 1, right});
        }
    }

    void prepare(size_t n) {
        for (int d = 0; d < 3; ++d) bins[d].assign(kBins, Bin());
        right_bounds.resize(std::max<size_t>(n, kBins) - 1);     // sbvh.h:124
    }
};

// A finished subtree: nodes with local child indices (root = 0) and its leaf ids in creation order
// (leaf `start` values are offsets into that local list).
struct Subtree {
    std::vector<BuildNode> nodes;
    std::vector<int32_t> leaf_ids;
};

struct Shared {
    const crt_triangle* tris;
    const float3* verts;
    bool spatial_enabled;
    float min_overlap;
    size_t parallel_threshold;   // subtrees with more references than this may fork
};

// The result of building a node depends only on its Spec and on the order of its own references (the
// reference algorithm never looks below the tail it is splitting), so the two children of a split can
// be built concurrently from private copies of their ranges and stitched back in the order the
// sequential algorithm would have produced them: right subtree's leaves first, then the left's.
Subtree build_rec(const Shared& sh, const Spec& spec, std::vector<Ref>&& refs) {
    Builder b;
    b.tris = sh.tris; b.verts = sh.verts; b.spatial_enabled = sh.spatial_enabled; b.min_overlap = sh.min_overlap;
    b.refs = std::move(refs);
    b.prepare(b.refs.size());
    Subtree out;
    if ((size_t)spec.n <= sh.parallel_threshold || spec.n <= kLeafRefs) {
        b.run(spec);
        out.nodes = std::move(b.nodes);
        out.leaf_ids = std::move(b.leaf_ids);
        return out;
    }
    Spec left, right;
    b.split_once(spec, left, right);
    std::vector<Ref> lrefs(b.refs.begin(), b.refs.begin() + left.n);
    std::vector<Ref> rrefs(b.refs.begin() + left.n, b.refs.end());
    std::vector<Ref>().swap(b.refs);
    std::future<Subtree> rf = std::async(std::launch::async, [&sh, right, &rrefs]() { return build_rec(sh, right, std::move(rrefs)); });
    Subtree L = build_rec(sh, left, std::move(lrefs));
    Subtree R = rf.get();

    BuildNode root;
    root.box = spec.box;
    const int r_base = 1, l_base = 1 + (int)R.nodes.size();
    root.child[0] = l_base;
    root.child[1] = r_base;
    out.nodes.reserve(1 + R.nodes.size() + L.nodes.size());
    out.nodes.push_back(root);
    for (BuildNode n : R.nodes) {
        if (!n.leaf) { n.child[0] += r_base; n.child[1] += r_base; }
        out.nodes.push_back(n);
    }
    const uint32_t l_leaf_base = (uint32_t)R.leaf_ids.size();
    for (BuildNode n : L.nodes) {
        if (!n.leaf) { n.child[0] += l_base; n.child[1] += l_base; }
        else n.start += l_leaf_base;
        out.nodes.push_back(n);
    }
    out.leaf_ids = std::move(R.leaf_ids);
    out.leaf_ids.insert(out.leaf_ids.end(), L.leaf_ids.begin(), L.leaf_ids.end());
    return out;
}

// CPUs this process may use: affinity mask and cgroup quota, overridable with CRT_BUILD_THREADS.
unsigned usable_threads() {
    if (const char* e = std::getenv("CRT_BUILD_THREADS")) {
        int v = std::atoi(e);
        if (v >= 1) return (unsigned)v;
    }
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)CPU_COUNT(&set));
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32];
        long period = 0;
        if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && quota[0] != 'm' && period > 0) {
            long q = std::atol(quota) / period;
            if (q >= 1) n = std::min<unsigned>(n, (unsigned)q);
        }
        std::fclose(f);
    }
    return std::max(1u, std::min(n, 64u));
}

}  // namespace

void SBVH::build(const crt_triangle* trs, size_t n_trs, const float3* vertices, size_t /*n_vertices*/, uint32_t flags) {
    flat_nodes.clear();
    triangle_indices.clear();
    triangles.clear();
    depth = 0;
    if (n_trs == 0) return;

    Spec root;
    root.n = (int)n_trs;
    std::vector<Ref> refs(n_trs);
    for (size_t i = 0; i < n_trs; ++i) {             // sbvh.h:109-118
        refs[i].id = (int)i;
        for (int j = 0; j < 3; ++j) refs[i].box.grow(vertices[trs[i].v[j]]);
        root.box.grow(refs[i].box);
    }
    Shared sh;
    sh.tris = trs;
    sh.verts = vertices;
    sh.spatial_enabled = !(flags & NO_SPATIAL_SPLITS);
    sh.min_overlap = root.box.half_area() * kSplitAlpha;           // sbvh.h:120
    const unsigned threads = usable_threads();
    // fork while a subtree holds more than ~1/(4*threads) of the input; below that, or with one
    // thread, the plain sequential algorithm runs
    sh.parallel_threshold = threads <= 1 ? n_trs : std::max<size_t>(4096, n_trs / (4 * (size_t)threads));
    Subtree tree = build_rec(sh, root, std::move(refs));
    struct { std::vector<BuildNode> nodes; std::vector<int32_t> leaf_ids; } b{std::move(tree.nodes), std::move(tree.leaf_ids)};

    // sbvh.h:130-139: the triangle array is re-ordered into leaf order (with duplicates).
    triangle_indices = b.leaf_ids;
    triangles.resize(triangle_indices.size());
    for (size_t i = 0; i < triangles.size(); ++i) triangles[i] = trs[triangle_indices[i]];

    // sbvh.h:285-324: every two-triangle leaf becomes an interior node over two single-
    // triangle leaves whose boxes are the FULL triangles' bounds (not the clipped refs').
    std::vector<BuildNode>& nodes = b.nodes;
    const size_t n_before = nodes.size();
    for (size_t i = 0; i < n_before; ++i) {
        if (!(nodes[i].leaf && nodes[i].range == 2)) continue;
        const uint32_t s = nodes[i].start;
        BuildNode kids[2];
        for (int k = 0; k < 2; ++k) {
            kids[k].leaf = true;
            kids[k].start = s + k;
            kids[k].range = 1;
            for (int j = 0; j < 3; ++j) kids[k].box.grow(vertices[triangles[s + k].v[j]]);
        }
        nodes[i].leaf = false;
        nodes[i].child[0] = (int)nodes.size();
        nodes[i].child[1] = (int)nodes.size() + 1;
        nodes.push_back(kids[0]);
        nodes.push_back(kids[1]);
    }

    // sbvh.h:570-609: breadth-first flattening; an interior node's children are adjacent.
    flat_nodes.reserve(nodes.size());
    std::deque<std::pair<int, int>> queue;   // (node, level)
    queue.emplace_back(0, 0);
    int next_child = 0;
    while (!queue.empty()) {
        auto [idx, level] = queue.front();
        queue.pop_front();
        const BuildNode& n = nodes[idx];
        crt_flatnode f;
        f.bmin[0] = n.box.lo.x; f.bmin[1] = n.box.lo.y; f.bmin[2] = n.box.lo.z;
        f.bmax[0] = n.box.hi.x; f.bmax[1] = n.box.hi.y; f.bmax[2] = n.box.hi.z;
        if (n.leaf) {
            f.bmin[3] = (float)n.start;
            f.bmax[3] = (float)n.range;
            depth = std::max(depth, level);
        } else {
            f.bmin[3] = (float)(next_child + 1);
            f.bmax[3] = 0.0f;
            next_child += 2;
            queue.emplace_back(n.child[0], level + 1);
            queue.emplace_back(n.child[1], level + 1);
        }
        flat_nodes.push_back(f);
    }
}

int SBVH::count_leaf() const {
    int c = 0;
    for (const crt_flatnode& n : flat_nodes) c += (n.bmax[3] != 0.0f);
    return c;
}

}  // namespace crt
