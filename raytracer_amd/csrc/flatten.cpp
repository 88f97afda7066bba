/*
 * flatten.cpp -- upload-time repacking of the host scene into the device layout.
 *
 * Input: the reference's host records (reference: src/scene.h:14-36) and its octree
 * (reference: src/octree.h:15-34, built by src/octree.c:384-419).  Output: the arrays
 * described in device_layout.h.  Pure host code; compiled with -ffp-contract=off because
 * the triangle normal and plane offset computed here must carry the same bits as the
 * reference's per-test recomputation (reference: src/raytracer.c:119-121,127).
 */
#include "flatten.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <cstring>
#include <limits>

namespace {

struct Box { float lo[3], hi[3]; };

inline Box empty_box() {
    Box b;
    for (int a = 0; a < 3; a++) { b.lo[a] = std::numeric_limits<float>::max(); b.hi[a] = -std::numeric_limits<float>::max(); }
    return b;
}
inline bool is_empty(const Box& b) { return b.lo[0] > b.hi[0] || b.lo[1] > b.hi[1] || b.lo[2] > b.hi[2]; }
inline void grow(Box& into, const Box& other) {
    if (is_empty(other)) return;
    for (int a = 0; a < 3; a++) { into.lo[a] = std::min(into.lo[a], other.lo[a]); into.hi[a] = std::max(into.hi[a], other.hi[a]); }
}
/* Intersection of a primitive's bounds with its cell; a primitive the builder's overlap test
 * let through by rounding but whose bounds miss the cell contributes nothing. */
inline Box clip(const Box& prim, const Box& cell) {
    Box r;
    for (int a = 0; a < 3; a++) { r.lo[a] = std::max(prim.lo[a], cell.lo[a]); r.hi[a] = std::min(prim.hi[a], cell.hi[a]); }
    return is_empty(r) ? empty_box() : r;
}

inline Box cell_of(const OctreeNode& n) {
    Box b;
    b.lo[0] = n.boundingBox.bottomLeftFrontCorner.x; b.lo[1] = n.boundingBox.bottomLeftFrontCorner.y; b.lo[2] = n.boundingBox.bottomLeftFrontCorner.z;
    b.hi[0] = n.boundingBox.topRightBackCorner.x; b.hi[1] = n.boundingBox.topRightBackCorner.y; b.hi[2] = n.boundingBox.topRightBackCorner.z;
    return b;
}

inline Box bounds_of(const Triangle& t) {
    Box b;
    const float* v[3] = {&t.v0.x, &t.v1.x, &t.v2.x};
    for (int a = 0; a < 3; a++) {
        b.lo[a] = std::min(v[0][a], std::min(v[1][a], v[2][a]));
        b.hi[a] = std::max(v[0][a], std::max(v[1][a], v[2][a]));
    }
    return b;
}

/* grow: added to the squared radius.  The reference's sphere test (reference: src/raytracer.c:88-116) takes the
 * difference of two products as large as the squared distance D^2 between ray origin and centre, so in fp32 it reports
 * hits for rays that pass the sphere at up to sqrt(r^2 + ~4e-7 D^2) from its centre; the boxes have to reach every ray
 * for which that arithmetic can say "hit", not just the geometric sphere. */
inline Box bounds_of(const Sphere& s, float grow = 0.0f) {
    Box b;
    const float* c = &s.position.x;
    float r = std::sqrt(s.radius * s.radius + grow);
    for (int a = 0; a < 3; a++) { b.lo[a] = c[a] - r; b.hi[a] = c[a] + r; }
    return b;
}

inline bool is_leaf(const OctreeNode& n) { return n.childNodeIndexes[0] == NODE_INDEX_UNDEF; }

/* reference: src/utils/vec3.c:56-70 -- length by sqrtf, three divisions, zero stays zero. */
inline void unit(float v[3]) {
    float len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (len != 0) { v[0] /= len; v[1] /= len; v[2] /= len; }
}

/* ---- refinement of oversized leaves -------------------------------------------------
 * The reference stops splitting a node as soon as a child keeps every primitive of its
 * parent (reference: src/octree.c:265), so a leaf can hold hundreds of primitives; a ray that
 * opens one would test them one after another while the other 63 lanes of its wavefront
 * wait.  Such a leaf is re-packed here as a small subtree in the same WideNode format: its
 * primitives are partitioned (not duplicated) into up to eight groups by three rounds of
 * median splits of their box centres, recursively, until a group has at most
 * RT_LEAF_TARGET primitives.  Each primitive is represented by its bounds clipped to the
 * leaf's cell, which is the only region this leaf is responsible for.  The set of
 * primitives a ray can reach is unchanged, so hit results are unchanged. */
#ifndef RT_LEAF_TARGET
#define RT_LEAF_TARGET 8u
#endif
#ifndef RT_PAD_FACTOR
#define RT_PAD_FACTOR 1.0e-5f
#endif

struct Prim { Box box; uint32_t index; bool sphere; };

struct Refiner {
    std::vector<WideNode>& nodes;
    std::vector<uint32_t>& leafIndex;
    float pad;
    uint32_t deepest = 0;   /* extra inner levels added below a reference leaf */
    uint32_t leaves = 0;
};

inline Box bounds_of(const Prim* p, size_t n) {
    Box b = empty_box();
    for (size_t i = 0; i < n; i++) grow(b, p[i].box);
    return b;
}

/* Appends the run for `n` primitives (spheres first) and returns (link, kind). */
inline void emit_run(Refiner& r, Prim* p, size_t n, uint32_t& link, uint32_t& kind) {
    std::stable_partition(p, p + n, [](const Prim& q) { return q.sphere; });
    uint32_t spheres = 0;
    link = (uint32_t)r.leafIndex.size();
    for (size_t i = 0; i < n; i++) { r.leafIndex.push_back(p[i].index); spheres += p[i].sphere ? 1u : 0u; }
    kind = (spheres << 20) | (uint32_t)(n - spheres);
    r.leaves++;
}

inline void split_median(Prim* p, size_t n) {
    Box centres = empty_box();
    for (size_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            float c = 0.5f * (p[i].box.lo[a] + p[i].box.hi[a]);
            centres.lo[a] = std::min(centres.lo[a], c);
            centres.hi[a] = std::max(centres.hi[a], c);
        }
    int axis = 0;
    float widest = -1.0f;
    for (int a = 0; a < 3; a++) if (centres.hi[a] - centres.lo[a] > widest) { widest = centres.hi[a] - centres.lo[a]; axis = a; }
    std::nth_element(p, p + n / 2, p + n, [axis](const Prim& x, const Prim& y) {
        float cx = x.box.lo[axis] + x.box.hi[axis], cy = y.box.lo[axis] + y.box.hi[axis];
        return cx < cy || (cx == cy && x.index < y.index);
    });
}

/* Builds the subtree for p[0..n) (n > RT_LEAF_TARGET) and returns its wide-node index. */
uint32_t refine(Refiner& r, Prim* p, size_t n, uint32_t level) {
    r.deepest = std::max(r.deepest, level + 1);
    const uint32_t self = (uint32_t)r.nodes.size();
    r.nodes.emplace_back();
    /* three rounds of median splits: 8 groups [bound[g], bound[g+1]) */
    size_t bound[9] = {0, 0, 0, 0, 0, 0, 0, 0, n};
    bound[4] = n / 2;
    split_median(p, n);
    bound[2] = bound[4] / 2; bound[6] = bound[4] + (n - bound[4]) / 2;
    split_median(p, bound[4]);
    split_median(p + bound[4], n - bound[4]);
    for (int h = 0; h < 4; h++) {
        size_t lo = bound[2 * h], hi = bound[2 * h + 2];
        bound[2 * h + 1] = lo + (hi - lo) / 2;
        if (hi - lo > 1) split_median(p + lo, hi - lo);
    }
    for (int c = 0; c < 8; c++) {
        const size_t lo = bound[c], cnt = bound[c + 1] - bound[c];
        uint32_t link = 0, kind = 0;
        Box b = empty_box();
        if (cnt > 0) {
            b = bounds_of(p + lo, cnt);
            if (cnt > RT_LEAF_TARGET && level + 3 < HIP_TRACE_MAX_TREE_LEVELS) {
                link = refine(r, p + lo, cnt, level + 1);
                kind = WIDE_INNER;
            } else {
                emit_run(r, p + lo, cnt, link, kind);
            }
        }
        WideNode& w = r.nodes[self];   /* re-fetch: the vector may have grown */
        const bool empty = is_empty(b);
        for (int a = 0; a < 3; a++) {
            w.lo[a][c] = empty ? std::numeric_limits<float>::max() : b.lo[a] - r.pad;
            w.hi[a][c] = empty ? -std::numeric_limits<float>::max() : b.hi[a] + r.pad;
        }
        w.ref[c].link = link;
        w.ref[c].kind = kind;
    }
    return self;
}


/* ---- private 8-wide bounding volume hierarchy -----------------------------------------
 * The reference's octree stores a primitive in every leaf cell it touches (reference:
 * src/octree.c:228-262), so a mesh of small triangles is referenced dozens of times and a ray
 * descends one level per halving of the scene.  Nothing in the trace path's results depends on
 * that topology: a walk only has to reach every primitive whose exact test can succeed.  This
 * builder produces, from the scene's spheres and triangles alone, a hierarchy in the same
 * WideNode format with each primitive stored once: a binary tree by binned surface-area
 * heuristic splits, then collapsed so that every node keeps up to eight children. */
#ifndef RT_BVH_LEAF
#define RT_BVH_LEAF 4u
#endif
#ifndef RT_BVH_MERGE
#define RT_BVH_MERGE 0u      /* a subtree of at most this many primitives becomes one leaf instead of a node */
#endif
#define RT_BVH_BINS 16

struct BinNode {
    Box box;
    uint32_t first, count;   /* the node's primitives: p[first, first + count) */
    uint32_t left, right;    /* children, or 0 for a leaf */
};

inline float half_area(const Box& b) {
    if (is_empty(b)) return 0.0f;
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

inline float centre(const Prim& q, int a) { return 0.5f * (q.box.lo[a] + q.box.hi[a]); }

/* Runs fn(begin, end, worker) over [0, count) cut into one contiguous piece per worker. */
template <typename F>
void for_pieces(uint32_t count, unsigned workers, F fn) {
    if (workers <= 1 || count < 262144u) { fn(0u, count, 0u); return; }
    std::vector<std::thread> pool;
    const uint32_t piece = (count + workers - 1) / workers;
    for (unsigned w = 0; w < workers; w++) {
        const uint32_t begin = std::min(count, w * piece), end = std::min(count, begin + piece);
        pool.emplace_back([=] { fn(begin, end, w); });
    }
    for (std::thread& t : pool) t.join();
}

struct BvhBuilder {
    std::vector<BinNode> bin;
    Prim* p = nullptr;
    unsigned workers = 1;      /* threads for the passes over one large range */
    uint32_t cutoff = 0;       /* a range of at most this many primitives is left to a later task (0: never) */
    struct Deferred { uint32_t node, first, count, depth; };
    std::vector<Deferred> deferred;

    struct Bins {
        Box box[3][RT_BVH_BINS];
        uint32_t n[3][RT_BVH_BINS];
        void reset() { for (int a = 0; a < 3; a++) for (int b = 0; b < RT_BVH_BINS; b++) { box[a][b] = empty_box(); n[a][b] = 0; } }
    };
    enum { MAX_WORKERS = 16 };

    uint32_t build(uint32_t first, uint32_t count, uint32_t depth) {
        const uint32_t self = (uint32_t)bin.size();
        bin.emplace_back();
        bin[self].first = first;
        bin[self].count = count;
        bin[self].left = bin[self].right = 0;
        if (cutoff && count <= cutoff && count > RT_BVH_LEAF) {
            bin[self].box = empty_box();             /* filled in when the task's subtree is merged */
            deferred.push_back(Deferred{self, first, count, depth});
            return self;
        }
        /* bounds of the boxes and of their centres, one pass */
        const unsigned lanes = std::min<unsigned>(MAX_WORKERS, std::max(1u, workers));
        Box partBox[MAX_WORKERS], partCentre[MAX_WORKERS];
        for (unsigned w = 0; w < lanes; w++) partBox[w] = partCentre[w] = empty_box();
        for_pieces(count, lanes, [&](uint32_t begin, uint32_t end, unsigned w) {
            Box bb = empty_box(), cc = empty_box();
            for (uint32_t i = begin; i < end; i++) {
                const Prim& q = p[first + i];
                grow(bb, q.box);
                for (int a = 0; a < 3; a++) {
                    float c = centre(q, a);
                    cc.lo[a] = std::min(cc.lo[a], c);
                    cc.hi[a] = std::max(cc.hi[a], c);
                }
            }
            partBox[w] = bb; partCentre[w] = cc;
        });
        Box box = empty_box(), cb = empty_box();
        for (unsigned w = 0; w < lanes; w++) { grow(box, partBox[w]); grow(cb, partCentre[w]); }
        bin[self].box = box;
        if (count <= RT_BVH_LEAF) return self;

        uint32_t mid = 0;
        int bestAxis = -1, bestBin = 0;
        float bestCost = std::numeric_limits<float>::max();
        float scaleOf[3];
        for (int a = 0; a < 3; a++) {
            const float extent = cb.hi[a] - cb.lo[a];
            scaleOf[a] = extent > 0.0f ? (float)RT_BVH_BINS / extent : 0.0f;
        }
        if (depth < 64) {
            Bins part[MAX_WORKERS];
            const unsigned used = count >= 262144u ? lanes : 1u;   /* for_pieces runs small ranges on one worker */
            for (unsigned w = 0; w < used; w++) part[w].reset();
            for_pieces(count, lanes, [&](uint32_t begin, uint32_t end, unsigned w) {
                Bins& mine = part[w];
                for (uint32_t i = begin; i < end; i++) {
                    const Prim& q = p[first + i];
                    for (int a = 0; a < 3; a++) {
                        int b = (int)((centre(q, a) - cb.lo[a]) * scaleOf[a]);
                        b = std::max(0, std::min(RT_BVH_BINS - 1, b));
                        grow(mine.box[a][b], q.box);
                        mine.n[a][b]++;
                    }
                }
            });
            for (unsigned w = 1; w < used; w++)
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < RT_BVH_BINS; b++) { grow(part[0].box[a][b], part[w].box[a][b]); part[0].n[a][b] += part[w].n[a][b]; }
            const Bins& all = part[0];
            for (int a = 0; a < 3; a++) {
                if (!(scaleOf[a] > 0.0f)) continue;
                float rightArea[RT_BVH_BINS];
                Box acc = empty_box();
                for (int b = RT_BVH_BINS - 1; b > 0; b--) { grow(acc, all.box[a][b]); rightArea[b] = half_area(acc); }
                acc = empty_box();
                uint32_t nLeft = 0;
                for (int b = 0; b + 1 < RT_BVH_BINS; b++) {
                    grow(acc, all.box[a][b]);
                    nLeft += all.n[a][b];
                    if (nLeft == 0 || nLeft == count) continue;
                    float cost = half_area(acc) * (float)nLeft + rightArea[b + 1] * (float)(count - nLeft);
                    if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = b; }
                }
            }
        }
        if (bestAxis >= 0) {
            const int a = bestAxis;
            const float k = scaleOf[a];
            const float lo = cb.lo[a];
            Prim* m = std::partition(p + first, p + first + count, [=](const Prim& q) {
                int b = (int)((centre(q, a) - lo) * k);
                b = std::max(0, std::min(RT_BVH_BINS - 1, b));
                return b <= bestBin;
            });
            mid = (uint32_t)(m - (p + first));
        }
        if (mid == 0 || mid == count) {   /* coincident centres, or too deep: halve by rank */
            mid = count / 2;
            split_median(p + first, count);
        }
        uint32_t l = build(first, mid, depth + 1);
        uint32_t r = build(first + mid, count - mid, depth + 1);
        bin[self].left = l;
        bin[self].right = r;
        return self;
    }

    /* Builds every deferred range on `threads` threads and splices the subtrees in.  The result
     * does not depend on the number of threads: each range is built by the same sequential code
     * and the subtrees are appended in the order the ranges were deferred. */
    void finish(unsigned threads) {
        std::vector<BvhBuilder> sub(deferred.size());
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t i = next++; i < deferred.size(); i = next++) {
                sub[i].p = p;
                sub[i].bin.reserve(deferred[i].count);
                sub[i].build(deferred[i].first, deferred[i].count, deferred[i].depth);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
        work();
        for (std::thread& t : pool) t.join();
        for (size_t i = 0; i < deferred.size(); i++) {
            const uint32_t base = (uint32_t)bin.size();     /* local node j >= 1 lands at base + j - 1 */
            const std::vector<BinNode>& nodes = sub[i].bin;
            auto moved = [&](BinNode n) {
                if (n.left || n.right) { n.left += base - 1; n.right += base - 1; }
                return n;
            };
            bin[deferred[i].node] = moved(nodes[0]);
            for (size_t j = 1; j < nodes.size(); j++) bin.push_back(moved(nodes[j]));
            sub[i].bin = std::vector<BinNode>();
        }
        deferred.clear();
    }
};

/* Slot s of a wide node is visited k-th by a ray whose direction-sign mask is s ^ k (bit0: x,
 * bit1: z, bit2: y; a set bit in s is the + side).  Children are therefore placed in the slot whose
 * corner direction best matches their offset from the node's centre, so that the fixed visiting
 * order is roughly front to back for every ray direction. */
void assign_slots(const Box& parent, const Box* child, int n, int slotOf[8]) {
    float pc[3];
    for (int a = 0; a < 3; a++) pc[a] = 0.5f * (parent.lo[a] + parent.hi[a]);
    float score[8][8];
    for (int i = 0; i < n; i++) {
        float d[3];
        for (int a = 0; a < 3; a++) d[a] = 0.5f * (child[i].lo[a] + child[i].hi[a]) - pc[a];
        for (int s = 0; s < 8; s++)
            score[i][s] = ((s & 1) ? d[0] : -d[0]) + ((s & 2) ? d[2] : -d[2]) + ((s & 4) ? d[1] : -d[1]);
    }
    bool childDone[8] = {false}, slotDone[8] = {false};
    for (int round = 0; round < n; round++) {
        int bi = -1, bs = -1;
        float best = -std::numeric_limits<float>::max();
        for (int i = 0; i < n; i++) {
            if (childDone[i]) continue;
            for (int s = 0; s < 8; s++)
                if (!slotDone[s] && score[i][s] > best) { best = score[i][s]; bi = i; bs = s; }
        }
        childDone[bi] = true;
        slotDone[bs] = true;
        slotOf[bi] = bs;
    }
}

uint32_t collapse(const BvhBuilder& b, Refiner& r, uint32_t binIndex, uint32_t level) {
    r.deepest = std::max(r.deepest, level + 1);
    const uint32_t self = (uint32_t)r.nodes.size();
    r.nodes.emplace_back();
    uint32_t kids[8];
    int n = 0;
    const BinNode& top = b.bin[binIndex];
    if (top.left == 0 && top.right == 0) {
        kids[n++] = binIndex;                 /* a root that is a single leaf */
    } else {
        kids[n++] = top.left;
        kids[n++] = top.right;
        while (n < 8) {
            int open = -1;
            float widest = -1.0f;
            for (int i = 0; i < n; i++) {
                const BinNode& k = b.bin[kids[i]];
                if (k.left == 0 && k.right == 0) continue;
                float area = half_area(k.box);
                if (area > widest) { widest = area; open = i; }
            }
            if (open < 0) break;
            const BinNode& k = b.bin[kids[open]];
            kids[open] = k.left;
            kids[n++] = k.right;
        }
    }
    Box boxes[8];
    for (int i = 0; i < n; i++) boxes[i] = b.bin[kids[i]].box;
    int slotOf[8];
    assign_slots(top.box, boxes, n, slotOf);
    bool used[8] = {false};
    for (int i = 0; i < n; i++) {
        const int c = slotOf[i];
        used[c] = true;
        const BinNode k = b.bin[kids[i]];
        uint32_t link = 0, kind = 0;
        if ((k.left == 0 && k.right == 0) || k.count <= RT_BVH_MERGE) {
            emit_run(r, b.p + k.first, k.count, link, kind);
        } else if (level + 8 >= HIP_TRACE_MAX_TREE_LEVELS) {
            link = refine(r, b.p + k.first, k.count, level + 1);   /* balanced below here */
            kind = WIDE_INNER;
        } else {
            link = collapse(b, r, kids[i], level + 1);
            kind = WIDE_INNER;
        }
        WideNode& w = r.nodes[self];
        const bool empty = is_empty(k.box);
        for (int a = 0; a < 3; a++) {
            w.lo[a][c] = empty ? std::numeric_limits<float>::max() : k.box.lo[a] - r.pad;
            w.hi[a][c] = empty ? -std::numeric_limits<float>::max() : k.box.hi[a] + r.pad;
        }
        w.ref[c].link = link;
        w.ref[c].kind = kind;
    }
    WideNode& w = r.nodes[self];
    for (int c = 0; c < 8; c++) {
        if (used[c]) continue;
        for (int a = 0; a < 3; a++) { w.lo[a][c] = std::numeric_limits<float>::max(); w.hi[a][c] = -std::numeric_limits<float>::max(); }
        w.ref[c].link = 0;
        w.ref[c].kind = 0;
    }
    return self;
}

}  // namespace

static int quantise_and_pack(const Scene* scene, FlatScene& out, std::string& error);

/* The hierarchy built here from the primitives alone; the caller's octree is not consulted. */
static int build_private_hierarchy(const Scene* scene, FlatScene& out, std::string& error) {
    std::vector<Prim> prims;
    prims.reserve((size_t)scene->sphereCount + scene->triangleCount);
    float scale = 32.0f;
    /* sphere boxes cover the reference arithmetic's verdicts for ray origins up to RT_SPHERE_REACH scene sizes from the
     * origin of coordinates (D <= RT_SPHERE_DISTANCE * size): beyond that, a frame widens all boxes instead */
    float size = 32.0f;
    for (uint32_t i = 0; i < scene->sphereCount; i++) {
        const Box raw = bounds_of(scene->spheres[i]);
        for (int a = 0; a < 3; a++)
            if (std::isfinite(raw.lo[a]) && std::isfinite(raw.hi[a])) size = std::max(size, std::max(std::fabs(raw.lo[a]), std::fabs(raw.hi[a])));
    }
    for (uint32_t i = 0; i < scene->triangleCount; i++) {
        const float* v = &scene->triangles[i].v0.x;
        for (int k = 0; k < 9; k++) if (std::isfinite(v[k])) size = std::max(size, std::fabs(v[k]));
    }
    const float sphereGrow = 4e-7f * (RT_SPHERE_DISTANCE * size) * (RT_SPHERE_DISTANCE * size);
    out.sphereGrow = sphereGrow;
    for (uint32_t i = 0; i < scene->sphereCount; i++) prims.push_back(Prim{bounds_of(scene->spheres[i], sphereGrow), i, true});
    for (uint32_t i = 0; i < scene->triangleCount; i++) prims.push_back(Prim{bounds_of(scene->triangles[i]), i, false});
    /* min/max drop a NaN, so finiteness is checked on the coordinates themselves */
    for (uint32_t i = 0; i < scene->sphereCount; i++) {
        const Sphere& sp = scene->spheres[i];
        if (!std::isfinite(sp.position.x) || !std::isfinite(sp.position.y) || !std::isfinite(sp.position.z) || !std::isfinite(sp.radius)) {
            error = "a sphere has a coordinate or radius that is not finite";
            return -1;
        }
    }
    for (uint32_t i = 0; i < scene->triangleCount; i++) {
        const float* v = &scene->triangles[i].v0.x;   /* v0, v1, v2: nine consecutive floats */
        for (int k = 0; k < 9; k++)
            if (!std::isfinite(v[k])) { error = "a triangle has a coordinate that is not finite"; return -1; }
    }
    for (const Prim& q : prims)
        for (int a = 0; a < 3; a++) {
            if (!std::isfinite(q.box.lo[a]) || !std::isfinite(q.box.hi[a])) { error = "a sphere's bounds overflow"; return -1; }
            scale = std::max(scale, std::max(std::fabs(q.box.lo[a]), std::fabs(q.box.hi[a])));
        }
    out.extent = scale;
    Refiner refiner{out.nodes, out.leafIndex, scale * RT_PAD_FACTOR};
    if (prims.empty()) {
        out.nodes.emplace_back();
        WideNode& w = out.nodes[0];
        for (int c = 0; c < 8; c++) {
            for (int a = 0; a < 3; a++) { w.lo[a][c] = std::numeric_limits<float>::max(); w.hi[a][c] = -std::numeric_limits<float>::max(); }
            w.ref[c].link = 0;
            w.ref[c].kind = 0;
        }
        out.treeLevels = 1;
    } else {
        /* large scenes: the passes over the top ranges and the subtrees below them run on several
         * threads (RT_BUILD_THREADS overrides the count; the tree is the same for any count) */
        unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char* env = std::getenv("RT_BUILD_THREADS")) threads = (unsigned)std::max(1, std::atoi(env));
        if (prims.size() < 200000) threads = 1;
        BvhBuilder builder;
        builder.p = prims.data();
        builder.bin.reserve(prims.size());
        builder.workers = threads;
        builder.cutoff = threads > 1 ? std::max<uint32_t>(8192u, (uint32_t)(prims.size() / (threads * 8u))) : 0u;
        auto T0 = std::chrono::steady_clock::now();
        builder.build(0, (uint32_t)prims.size(), 0);
        auto T1 = std::chrono::steady_clock::now();
        builder.finish(threads);
        auto T2 = std::chrono::steady_clock::now();
        if (std::getenv("RT_BUILD_TIMING")) fprintf(stderr, "top %.3f s (%zu deferred), subtrees %.3f s\n", std::chrono::duration<double>(T1-T0).count(), builder.deferred.size(), std::chrono::duration<double>(T2-T1).count());
        collapse(builder, refiner, 0, 0);
        if (std::getenv("RT_BUILD_TIMING")) fprintf(stderr, "collapse %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now()-T2).count());
        out.treeLevels = std::max(1u, refiner.deepest);
    }
    out.leafCount = refiner.leaves;
    if (out.leafIndex.empty()) out.leafIndex.push_back(0);
    return quantise_and_pack(scene, out, error);
}

int rt_flatten(const Scene* scene, const Octree* tree, uint32_t accel, FlatScene& out, std::string& error) {
    out = FlatScene();
    if (!scene) { error = "scene is NULL"; return -1; }
    if (accel == HIP_TRACE_ACCEL_CALLER_OCTREE && !tree) { error = "octree is NULL"; return -1; }
    if (scene->materialCount == 0 || !scene->materials) { error = "scene has no materials (material 0 is the background)"; return -1; }
    if ((scene->planeCount && !scene->planes) || (scene->sphereCount && !scene->spheres) ||
        (scene->triangleCount && !scene->triangles) || (scene->pointLightCount && !scene->pointLights)) {
        error = "scene array pointer is NULL while its count is not zero";
        return -1;
    }
    const uint64_t primitives = (uint64_t)scene->planeCount + scene->sphereCount + scene->triangleCount;
    if (primitives >= 0xFFFFFFFFull) { error = "more than 2^32-2 primitives"; return -1; }
    for (uint32_t i = 0; i < scene->planeCount; i++)
        if (scene->planes[i].materialIndex >= scene->materialCount) { error = "plane material index out of range"; return -1; }
    for (uint32_t i = 0; i < scene->sphereCount; i++)
        if (scene->spheres[i].materialIndex >= scene->materialCount) { error = "sphere material index out of range"; return -1; }
    for (uint32_t i = 0; i < scene->triangleCount; i++)
        if (scene->triangles[i].materialIndex >= scene->materialCount) { error = "triangle material index out of range"; return -1; }
    if (accel != HIP_TRACE_ACCEL_CALLER_OCTREE) return build_private_hierarchy(scene, out, error);

    /* ---- octree: validate, measure depth ---- */
    const uint32_t nodeCount = tree->nodeCount;
    if (nodeCount == 0 || !tree->nodes) { error = "octree has no nodes"; return -5; }
    if (tree->indexCount && !tree->indexes) { error = "octree index array is NULL"; return -5; }
    std::vector<uint8_t> level(nodeCount, 0);
    std::vector<uint8_t> reached(nodeCount, 0);
    reached[0] = 1;
    uint32_t deepest = 0;
    for (uint32_t n = 0; n < nodeCount; n++) {
        const OctreeNode& node = tree->nodes[n];
        if (!reached[n]) continue;   /* stray record: never referenced, ignore */
        deepest = std::max<uint32_t>(deepest, level[n]);
        if (is_leaf(node)) {
            uint64_t sEnd = (uint64_t)node.sphereIndexOffset + node.sphereIndexCount;
            uint64_t tEnd = (uint64_t)node.triangleIndexOffset + node.triangleIndexCount;
            if (sEnd > tree->indexCount || tEnd > tree->indexCount) { error = "octree leaf run exceeds the index array"; return -5; }
            if (node.sphereIndexCount && node.triangleIndexCount && node.triangleIndexOffset != sEnd) {
                error = "octree leaf runs are not adjacent (spheres, then triangles)"; return -5;
            }
            if (node.sphereIndexCount > WIDE_MAX_LEAF_SPHERES || node.triangleIndexCount > WIDE_MAX_LEAF_TRIANGLES) {
                error = "octree leaf holds more primitives than the device layout encodes"; return -5;
            }
            for (uint32_t i = 0; i < node.sphereIndexCount; i++)
                if (tree->indexes[node.sphereIndexOffset + i] >= scene->sphereCount) { error = "octree sphere index out of range"; return -5; }
            for (uint32_t i = 0; i < node.triangleIndexCount; i++)
                if (tree->indexes[node.triangleIndexOffset + i] >= scene->triangleCount) { error = "octree triangle index out of range"; return -5; }
            continue;
        }
        if (level[n] + 1u >= HIP_TRACE_MAX_TREE_LEVELS) { error = "octree is deeper than HIP_TRACE_MAX_TREE_LEVELS"; return -5; }
        for (int c = 0; c < 8; c++) {
            int32_t child = node.childNodeIndexes[c];
            if (child <= (int32_t)n || (uint32_t)child >= nodeCount) { error = "octree child id is not a later node"; return -5; }
            if (reached[child]) { error = "octree node has two parents"; return -5; }
            reached[child] = 1;
            level[child] = (uint8_t)(level[n] + 1);
        }
    }

    /* ---- tight bounds, children before parents (children have larger ids) ---- */
    std::vector<Box> tight(nodeCount, empty_box());
    Box rootCell = cell_of(tree->nodes[0]);
    float scale = 32.0f;
    for (int a = 0; a < 3; a++) scale = std::max(scale, std::max(std::fabs(rootCell.lo[a]), std::fabs(rootCell.hi[a])));
    const float pad = scale * RT_PAD_FACTOR;
    out.extent = scale;
    for (uint32_t n = nodeCount; n-- > 0;) {
        if (!reached[n]) continue;
        const OctreeNode& node = tree->nodes[n];
        Box acc = empty_box();
        if (is_leaf(node)) {
            Box cell = cell_of(node);
            for (uint32_t i = 0; i < node.sphereIndexCount; i++)
                grow(acc, clip(bounds_of(scene->spheres[tree->indexes[node.sphereIndexOffset + i]]), cell));
            for (uint32_t i = 0; i < node.triangleIndexCount; i++)
                grow(acc, clip(bounds_of(scene->triangles[tree->indexes[node.triangleIndexOffset + i]]), cell));
        } else {
            for (int c = 0; c < 8; c++) grow(acc, tight[node.childNodeIndexes[c]]);
        }
        tight[n] = acc;
    }

    /* ---- wide nodes ---- */
    const bool rootIsLeaf = is_leaf(tree->nodes[0]);
    std::vector<uint32_t> wideOf(nodeCount, 0xFFFFFFFFu);
    uint32_t wideCount = 0;
    if (rootIsLeaf) {
        wideCount = 1;
    } else {
        for (uint32_t n = 0; n < nodeCount; n++)
            if (reached[n] && !is_leaf(tree->nodes[n])) wideOf[n] = wideCount++;
    }
    out.nodes.resize(wideCount);
    out.leafIndex.assign(tree->indexes, tree->indexes + tree->indexCount);
    Refiner refiner{out.nodes, out.leafIndex, pad};
    std::vector<Prim> scratch;
    auto fill_child = [&](uint32_t wideIndex, int c, int64_t childId) {
        WideNode& w = out.nodes[wideIndex];
        Box b = childId >= 0 ? tight[childId] : empty_box();
        bool empty = is_empty(b);
        for (int a = 0; a < 3; a++) {
            w.lo[a][c] = empty ? std::numeric_limits<float>::max() : b.lo[a] - pad;
            w.hi[a][c] = empty ? -std::numeric_limits<float>::max() : b.hi[a] + pad;
        }
        if (childId < 0) { w.ref[c].link = 0; w.ref[c].kind = 0; return; }
        const OctreeNode& child = tree->nodes[childId];
        if (is_leaf(child)) {
            if (child.sphereIndexCount + child.triangleIndexCount > RT_LEAF_TARGET && !empty) {
                Box cell = cell_of(child);
                scratch.clear();
                for (uint32_t i = 0; i < child.sphereIndexCount; i++) {
                    uint32_t k = tree->indexes[child.sphereIndexOffset + i];
                    Box pb = clip(bounds_of(scene->spheres[k]), cell);
                    if (!is_empty(pb)) scratch.push_back(Prim{pb, k, true});
                }
                for (uint32_t i = 0; i < child.triangleIndexCount; i++) {
                    uint32_t k = tree->indexes[child.triangleIndexOffset + i];
                    Box pb = clip(bounds_of(scene->triangles[k]), cell);
                    if (!is_empty(pb)) scratch.push_back(Prim{pb, k, false});
                }
                if (scratch.size() > RT_LEAF_TARGET) {
                    uint32_t sub = refine(refiner, scratch.data(), scratch.size(), level[childId]);
                    out.nodes[wideIndex].ref[c].link = sub;     /* refine() may have moved the vector */
                    out.nodes[wideIndex].ref[c].kind = WIDE_INNER;
                    return;
                }
            }
            w.ref[c].link = child.sphereIndexCount ? child.sphereIndexOffset : child.triangleIndexOffset;
            w.ref[c].kind = (child.sphereIndexCount << 20) | child.triangleIndexCount;
            if (!empty) out.leafCount++;
        } else {
            w.ref[c].link = wideOf[childId];
            w.ref[c].kind = WIDE_INNER;
        }
    };
    if (rootIsLeaf) {
        /* A tree that is one leaf (the default scene) gets a synthetic parent with one child. */
        fill_child(0, 0, 0);
        for (int c = 1; c < 8; c++) fill_child(0, c, -1);
    } else {
        for (uint32_t n = 0; n < nodeCount; n++) {
            if (wideOf[n] == 0xFFFFFFFFu) continue;
            for (int c = 0; c < 8; c++) fill_child(wideOf[n], c, tree->nodes[n].childNodeIndexes[c]);
        }
    }
    /* inner levels on the longest path, including levels added under refined leaves
     * (refine() records absolute levels: it is entered with the level of the leaf it replaces) */
    out.treeLevels = std::max(rootIsLeaf ? 1u : std::max(1u, deepest), refiner.deepest + (rootIsLeaf ? 1u : 0u));
    out.leafCount += refiner.leaves;
    if (out.leafIndex.empty()) out.leafIndex.push_back(0);
    return quantise_and_pack(scene, out, error);
}

/* Renumbers the wide nodes breadth first so that the inner children of every node are consecutive
 * nodes in child order (QNode::firstInner), whatever order they were built in. */
static void make_inner_children_consecutive(std::vector<WideNode>& nodes) {
    const uint32_t count = (uint32_t)nodes.size();
    if (count <= 1) return;
    std::vector<uint32_t> order;            /* order[new index] = old index */
    order.reserve(count);
    order.push_back(0);
    for (uint32_t at = 0; at < order.size(); at++) {
        const WideNode& w = nodes[order[at]];
        for (int c = 0; c < 8; c++)
            if (w.ref[c].kind & WIDE_INNER) order.push_back(w.ref[c].link);
    }
    /* nodes that nothing refers to (there are none in a well-formed build) keep a place at the end */
    std::vector<uint32_t> newIndex(count, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < order.size(); i++) newIndex[order[i]] = i;
    for (uint32_t i = 0; i < count; i++)
        if (newIndex[i] == 0xFFFFFFFFu) { newIndex[i] = (uint32_t)order.size(); order.push_back(i); }
    std::vector<WideNode> moved(count);
    for (uint32_t i = 0; i < count; i++) {
        WideNode w = nodes[order[i]];
        for (int c = 0; c < 8; c++)
            if (w.ref[c].kind & WIDE_INNER) w.ref[c].link = newIndex[w.ref[c].link];
        moved[i] = w;
    }
    nodes.swap(moved);
}

static int quantise_and_pack(const Scene* scene, FlatScene& out, std::string& error) {
    make_inner_children_consecutive(out.nodes);
    /* ---- quantise the wide nodes (device_layout.h: QNode) ---- */
    out.qnodes.resize(out.nodes.size());
    std::atomic<bool> tooFat{false};
    const unsigned packWorkers = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    for_pieces((uint32_t)out.nodes.size(), packWorkers, [&](uint32_t nodeBegin, uint32_t nodeEnd, unsigned) {
    for (size_t n = nodeBegin; n < nodeEnd; n++) {
        const WideNode& w = out.nodes[n];
        QNode& q = out.qnodes[n];
        std::memset(&q, 0, sizeof q);
        bool present[8];
        for (int c = 0; c < 8; c++) present[c] = w.lo[0][c] <= w.hi[0][c] && w.lo[1][c] <= w.hi[1][c] && w.lo[2][c] <= w.hi[2][c];
        for (int a = 0; a < 3; a++) {
            float lo = std::numeric_limits<float>::max(), hi = -std::numeric_limits<float>::max();
            for (int c = 0; c < 8; c++) if (present[c]) { lo = std::min(lo, w.lo[a][c]); hi = std::max(hi, w.hi[a][c]); }
            if (lo > hi) { lo = 0.0f; hi = 1.0f; }            /* node without children: any grid */
            /* a grid step that certainly spans [lo, hi] in 255 steps, and never zero */
            float step = (float)(((double)hi - (double)lo) / 255.0 * (1.0 + 1e-6));
            if (!(step > 0.0f)) step = std::numeric_limits<float>::min();
            q.origin[a] = lo;
            q.scale[a] = step;
            for (int c = 0; c < 8; c++) {
                if (!present[c]) { q.qlo[a][c] = 255; q.qhi[a][c] = 0; continue; }
                /* the kernel evaluates origin + step * q in fp32 (inside a fused expression); choose q
                 * against the exactly rounded fp32 product-sum with one grid step of slack */
                int ql = (int)std::floor(((double)w.lo[a][c] - (double)lo) / (double)step);
                int qh = (int)std::ceil(((double)w.hi[a][c] - (double)lo) / (double)step);
                ql = std::max(0, std::min(255, ql));
                qh = std::max(0, std::min(255, qh));
                while (ql > 0 && (double)lo + (double)step * ql > (double)w.lo[a][c]) ql--;
                while (qh < 255 && (double)lo + (double)step * qh < (double)w.hi[a][c]) qh++;
                q.qlo[a][c] = (uint8_t)ql;
                q.qhi[a][c] = (uint8_t)qh;
            }
        }
        for (int c = 0; c < 8; c++) {
            q.link[c] = w.ref[c].link;
            if (w.ref[c].kind & WIDE_INNER) {
                if (q.innerMask == 0) q.firstInner = w.ref[c].link;
                q.innerMask |= 1u << c;
                q.kind[c] = QNODE_INNER;
            } else {
                const uint32_t spheres = w.ref[c].kind >> 20, triangles = w.ref[c].kind & 0xFFFFFu;
                if (spheres > QNODE_MAX_LEAF_SPHERES || triangles > QNODE_MAX_LEAF_TRIANGLES) {
                    tooFat = true;
                    continue;
                }
                q.kind[c] = (uint16_t)((spheres << 11) | triangles);
            }
        }
    }
    });
    if (tooFat) { error = "a leaf holds more primitives than the device layout encodes"; return -5; }
    if (out.qnodes.size() > (1u << 24)) { error = "more than 2^24 wide nodes (a walk stack entry keeps 24 bits of node number)"; return -5; }

    /* ---- primitives ---- */
    out.trianglePlane.resize(std::max<uint32_t>(scene->triangleCount, 1));
    out.triangleCorner.resize((size_t)std::max<uint32_t>(scene->triangleCount, 1) * 3);
    out.triangleMaterial.resize(std::max<uint32_t>(scene->triangleCount, 1));
    for_pieces(scene->triangleCount, packWorkers, [&](uint32_t triBegin, uint32_t triEnd, unsigned) {
    for (uint32_t i = triBegin; i < triEnd; i++) {
        const Triangle& t = scene->triangles[i];
        /* reference: src/raytracer.c:119-121 -- normal = norm(cross(v1 - v0, v2 - v0)) */
        float e1[3] = {t.v1.x - t.v0.x, t.v1.y - t.v0.y, t.v1.z - t.v0.z};
        float e2[3] = {t.v2.x - t.v0.x, t.v2.y - t.v0.y, t.v2.z - t.v0.z};
        float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        unit(n);
        /* reference: src/raytracer.c:127 -- d = dot(normal, v0) */
        float d = n[0] * t.v0.x + n[1] * t.v0.y + n[2] * t.v0.z;
        out.trianglePlane[i] = Float4{n[0], n[1], n[2], d};
        out.triangleMaterial[i] = t.materialIndex;
        out.triangleCorner[3 * (size_t)i + 0] = Float4{t.v0.x, t.v0.y, t.v0.z, 0.0f};
        out.triangleCorner[3 * (size_t)i + 1] = Float4{t.v1.x, t.v1.y, t.v1.z, 0.0f};
        out.triangleCorner[3 * (size_t)i + 2] = Float4{t.v2.x, t.v2.y, t.v2.z, 0.0f};
    }
    });
    out.sphere.resize(std::max<uint32_t>(scene->sphereCount, 1));
    out.sphereMaterial.resize(std::max<uint32_t>(scene->sphereCount, 1));
    for (uint32_t i = 0; i < scene->sphereCount; i++) {
        const Sphere& s = scene->spheres[i];
        out.sphere[i] = Float4{s.position.x, s.position.y, s.position.z, s.radius};
        out.sphereMaterial[i] = s.materialIndex;
    }
    out.plane.resize(std::max<uint32_t>(scene->planeCount, 1));
    out.planeMaterial.resize(std::max<uint32_t>(scene->planeCount, 1));
    for (uint32_t i = 0; i < scene->planeCount; i++) {
        const Plane& p = scene->planes[i];
        out.plane[i] = Float4{p.normal.x, p.normal.y, p.normal.z, p.distanceFromOrigin};
        out.planeMaterial[i] = p.materialIndex;
    }
    out.material.resize((size_t)scene->materialCount * 3);
    for (uint32_t i = 0; i < scene->materialCount; i++) {
        const Material& m = scene->materials[i];
        out.material[3 * (size_t)i + 0] = Float4{m.reflectionIndex, m.refractionIndex, m.ambientWeight, m.diffuseWeight};
        out.material[3 * (size_t)i + 1] = Float4{m.color.r, m.color.g, m.color.b, m.specularExponent};
        out.material[3 * (size_t)i + 2] = Float4{m.specularWeight, 0.0f, 0.0f, 0.0f};
    }
    out.light.resize((size_t)std::max<uint32_t>(scene->pointLightCount, 1) * 2);
    for (uint32_t i = 0; i < scene->pointLightCount; i++) {
        const PointLight& l = scene->pointLights[i];
        out.light[2 * (size_t)i + 0] = Float4{l.position.x, l.position.y, l.position.z, l.strength};
        out.light[2 * (size_t)i + 1] = Float4{l.emissionColor.r, l.emissionColor.g, l.emissionColor.b, 0.0f};
    }
    for (uint32_t i = 0; i < scene->planeCount; i++)
        if (std::isfinite(scene->planes[i].distanceFromOrigin)) out.extent = std::max(out.extent, std::fabs(scene->planes[i].distanceFromOrigin));
    out.planeCount = scene->planeCount;
    out.sphereCount = scene->sphereCount;
    out.triangleCount = scene->triangleCount;
    out.lightCount = scene->pointLightCount;
    out.materialCount = scene->materialCount;
    return 0;
}

void rt_check_hierarchy(const Scene* scene, const FlatScene& flat, uint32_t accel, HipTraceHierarchyReport& rep) {
    std::memset(&rep, 0, sizeof rep);
    rep.wideNodeCount = (uint32_t)flat.qnodes.size();
    rep.leafCount = flat.leafCount;
    rep.leafIndexCount = (uint32_t)flat.leafIndex.size();
    rep.treeLevels = flat.treeLevels;
    rep.containmentChecked = accel == HIP_TRACE_ACCEL_PRIVATE_BVH ? 1u : 0u;
    uint32_t fnv = 2166136261u;
    auto mix = [&fnv](const void* data, size_t bytes) {
        const unsigned char* b = static_cast<const unsigned char*>(data);
        for (size_t i = 0; i < bytes; i++) { fnv ^= b[i]; fnv *= 16777619u; }
    };
    mix(flat.qnodes.data(), flat.qnodes.size() * sizeof(QNode));
    mix(flat.leafIndex.data(), flat.leafIndex.size() * sizeof(uint32_t));
    rep.checksum = fnv;
    std::vector<uint32_t> sphereSeen(scene->sphereCount, 0), triangleSeen(scene->triangleCount, 0);
    std::vector<uint8_t> nodeSeen(flat.qnodes.size(), 0);
    /* depth-first with the chain of enclosing child boxes (double precision, as wide as any fp32
     * evaluation of origin + scale * q can be: one grid step of slack is built into q) */
    struct Frame { uint32_t node; int next; };
    struct DBox { double lo[3], hi[3]; };
    std::vector<Frame> stack;
    std::vector<DBox> chain;
    if (flat.qnodes.empty()) { rep.badLinks = 1; return; }
    stack.push_back(Frame{0, 0});
    nodeSeen[0] = 1;
    while (!stack.empty()) {
        Frame& f = stack.back();
        rep.deepestLevel = std::max<uint32_t>(rep.deepestLevel, (uint32_t)stack.size());
        if (f.next == 8) { stack.pop_back(); if (!chain.empty()) chain.pop_back(); continue; }
        const int c = f.next++;
        const QNode& q = flat.qnodes[f.node];
        if (q.qlo[0][c] > q.qhi[0][c] || q.qlo[1][c] > q.qhi[1][c] || q.qlo[2][c] > q.qhi[2][c]) continue;
        rep.childCount++;
        DBox b;
        for (int a = 0; a < 3; a++) {
            b.lo[a] = (double)q.origin[a] + (double)q.scale[a] * q.qlo[a][c];
            b.hi[a] = (double)q.origin[a] + (double)q.scale[a] * q.qhi[a][c];
        }
        if (q.kind[c] & QNODE_INNER) {
            /* the walk derives an inner child's node from firstInner and innerMask, not from link[] */
            const uint32_t derived = q.firstInner + (uint32_t)__builtin_popcount(q.innerMask & ((1u << c) - 1u));
            if (!((q.innerMask >> c) & 1u) || derived != q.link[c]) { rep.badLinks++; continue; }
            if (q.link[c] >= flat.qnodes.size() || nodeSeen[q.link[c]] || stack.size() >= HIP_TRACE_MAX_TREE_LEVELS) { rep.badLinks++; continue; }
            nodeSeen[q.link[c]] = 1;
            chain.push_back(b);
            stack.push_back(Frame{q.link[c], 0});   /* invalidates f: not used again this turn */
            continue;
        }
        const uint32_t spheres = q.kind[c] >> 11, triangles = q.kind[c] & 0x7FFu;
        if ((q.innerMask >> c) & 1u) { rep.badLinks++; continue; }
        if ((uint64_t)q.link[c] + spheres + triangles > flat.leafIndex.size()) { rep.badLinks++; continue; }
        chain.push_back(b);
        for (uint32_t i = 0; i < spheres + triangles; i++) {
            const uint32_t k = flat.leafIndex[q.link[c] + i];
            const bool isSphere = i < spheres;
            if (k >= (isSphere ? scene->sphereCount : scene->triangleCount)) { rep.badLinks++; continue; }
            uint32_t& seen = isSphere ? sphereSeen[k] : triangleSeen[k];
            if (seen++) rep.duplicateReferences++;
            if (!rep.containmentChecked) continue;
            const Box pb = isSphere ? bounds_of(scene->spheres[k]) : bounds_of(scene->triangles[k]);
            for (const DBox& e : chain)
                for (int a = 0; a < 3; a++)
                    if ((double)pb.lo[a] < e.lo[a] || (double)pb.hi[a] > e.hi[a]) { rep.escapedPrimitives++; break; }
        }
        chain.pop_back();
    }
    for (uint32_t v : sphereSeen) rep.unreachedSpheres += v == 0;
    for (uint32_t v : triangleSeen) rep.unreachedTriangles += v == 0;
}
