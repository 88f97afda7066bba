/*
 * rt_octree.c -- octree builder over a scene's spheres and triangles.
 *
 * Stand-alone equivalent of the reference builder (reference:
 * src/octree.c:7-427).  For the same scene it yields the same tree: the same
 * node numbering (the eight children of a node are numbered together when the
 * node is split, before any of them is visited, src/octree.c:270-286), the same
 * boxes (including the child whose minimum corner keeps the parent's y,
 * src/octree.c:336-340), the same split rule (src/octree.c:265) and the same
 * leaf runs in indexes[] (spheres, then triangles, src/octree.c:365-378).
 * The overlap predicates evaluate the reference's fp32 expressions in its
 * operand order; terms that the reference multiplies by a literal zero are
 * dropped, which changes no comparison for finite coordinates.
 *
 * Unlike the reference the work lists live in LIFO arenas instead of a malloc
 * per node, inner nodes get zeroed offsets (the reference leaves them
 * unwritten), allocation failure returns NULL instead of crashing, and the
 * subtrees below the top levels are built on several threads (same bytes out).
 *
 * Compile with -ffp-contract=off.
 */
#include <float.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "rt_host.h"
#include "rt_vecmath.h"

typedef struct {
    Octree* tree;
    const Scene* scene;
    uint32_t* arena;      /* LIFO storage for the per-node "inside" lists */
    size_t arenaTop, arenaCapacity;
    int failed;
    int subtree;          /* node 0 of this build is not the scene's root */
} Builder;

static void rt_track(float value, float* lo, float* hi) {
    if (value < *lo) *lo = value;
    if (value > *hi) *hi = value;
}

static void rt_box_include(BoundingBox* box, Vec3 p) {
    if (p.x < box->bottomLeftFrontCorner.x) box->bottomLeftFrontCorner.x = p.x;
    if (p.y < box->bottomLeftFrontCorner.y) box->bottomLeftFrontCorner.y = p.y;
    if (p.z < box->bottomLeftFrontCorner.z) box->bottomLeftFrontCorner.z = p.z;
    if (p.x > box->topRightBackCorner.x) box->topRightBackCorner.x = p.x;
    if (p.y > box->topRightBackCorner.y) box->topRightBackCorner.y = p.y;
    if (p.z > box->topRightBackCorner.z) box->topRightBackCorner.z = p.z;
}

/* The root box starts as the single point (0,0,0), so it always contains the origin
 * (reference: src/octree.c:7-69). */
static BoundingBox rt_root_box(const Scene* scene) {
    BoundingBox box;
    memset(&box, 0, sizeof box);
    for (uint32_t i = 0; i < scene->sphereCount; i++) {
        const Sphere* s = &scene->spheres[i];
        float r = s->radius;
        rt_box_include(&box, v3_add(s->position, v3(0.0f, 0.0f, r)));
        rt_box_include(&box, v3_add(s->position, v3(0.0f, 0.0f, -r)));
        rt_box_include(&box, v3_add(s->position, v3(0.0f, r, 0.0f)));
        rt_box_include(&box, v3_add(s->position, v3(0.0f, -r, 0.0f)));
        rt_box_include(&box, v3_add(s->position, v3(-r, 0.0f, 0.0f)));
        rt_box_include(&box, v3_add(s->position, v3(r, 0.0f, 0.0f)));
    }
    for (uint32_t i = 0; i < scene->triangleCount; i++) {
        const Triangle* t = &scene->triangles[i];
        rt_box_include(&box, t->v0);
        rt_box_include(&box, t->v1);
        rt_box_include(&box, t->v2);
    }
    return box;
}

/* Squared-distance test between a sphere and a box (reference: src/octree.c:82-107). */
static int rt_sphere_touches_box(const Sphere* s, const BoundingBox* box) {
    const float* c = &s->position.x;
    const float* lo = &box->bottomLeftFrontCorner.x;
    const float* hi = &box->topRightBackCorner.x;
    float remaining = s->radius * s->radius;
    for (int axis = 0; axis < 3; axis++) {
        if (c[axis] < lo[axis]) {
            remaining -= (c[axis] - lo[axis]) * (c[axis] - lo[axis]);
        } else if (c[axis] > hi[axis]) {
            remaining -= (c[axis] - hi[axis]) * (c[axis] - hi[axis]);
        }
    }
    return remaining > 0;
}

/* Separating-axis test of a triangle against a box: 3 box axes, the triangle normal,
 * 9 edge x axis products (reference: src/octree.c:124-201). */
static int rt_triangle_touches_box(const Triangle* t, const BoundingBox* box) {
    const float lo[3] = {box->bottomLeftFrontCorner.x, box->bottomLeftFrontCorner.y, box->bottomLeftFrontCorner.z};
    const float hi[3] = {box->topRightBackCorner.x, box->topRightBackCorner.y, box->topRightBackCorner.z};
    const Vec3 vert[3] = {t->v0, t->v1, t->v2};

    for (int axis = 0; axis < 3; axis++) {
        float tmin = INFINITY, tmax = -INFINITY;
        for (int k = 0; k < 3; k++) rt_track((&vert[k].x)[axis], &tmin, &tmax);
        if (tmax < lo[axis] || tmin > hi[axis]) return 0;
    }

    Vec3 normal = v3_unit(v3_cross(v3_sub(t->v1, t->v0), v3_sub(t->v2, t->v0)));
    float offset = v3_dot(normal, t->v0);
    {
        float bmin = INFINITY, bmax = -INFINITY;
        for (int corner = 0; corner < 8; corner++) {
            Vec3 p = v3((corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2]);
            rt_track(v3_dot(normal, p), &bmin, &bmax);
        }
        if (bmax < offset || bmin > offset) return 0;
    }

    const Vec3 edge[3] = {v3_sub(t->v0, t->v1), v3_sub(t->v1, t->v2), v3_sub(t->v2, t->v0)};
    for (int i = 0; i < 3; i++) {
        const float e[3] = {edge[i].x, edge[i].y, edge[i].z};
        for (int j = 0; j < 3; j++) {
            /* edge x unit_j has two non-zero components: (+e[c] on axis b, -e[b] on axis c)
             * with (b, c) = the two axes after j in the order that makes the first product
             * the lower-numbered axis, matching left-to-right evaluation of the dot product. */
            int a0, a1; float k0, k1;
            if (j == 0)      { a0 = 1; k0 = e[2];  a1 = 2; k1 = -e[1]; }
            else if (j == 1) { a0 = 0; k0 = -e[2]; a1 = 2; k1 = e[0]; }
            else             { a0 = 0; k0 = e[1];  a1 = 1; k1 = -e[0]; }
            float bmin = INFINITY, bmax = -INFINITY, tmin = INFINITY, tmax = -INFINITY;
            rt_track(k0 * lo[a0] + k1 * lo[a1], &bmin, &bmax);
            rt_track(k0 * hi[a0] + k1 * lo[a1], &bmin, &bmax);
            rt_track(k0 * lo[a0] + k1 * hi[a1], &bmin, &bmax);
            rt_track(k0 * hi[a0] + k1 * hi[a1], &bmin, &bmax);
            for (int k = 0; k < 3; k++) {
                const float* p = &vert[k].x;
                rt_track(k0 * p[a0] + k1 * p[a1], &tmin, &tmax);
            }
            if (bmax < tmin || bmin > tmax) return 0;
        }
    }
    return 1;
}

static size_t rt_arena_reserve(Builder* b, size_t count) {
    if (b->arenaTop + count > b->arenaCapacity) {
        size_t grown = b->arenaCapacity * 2;
        if (grown < b->arenaTop + count) grown = b->arenaTop + count;
        uint32_t* moved = (uint32_t*)realloc(b->arena, grown * sizeof(uint32_t));
        if (!moved) { b->failed = 1; return (size_t)-1; }
        b->arena = moved;
        b->arenaCapacity = grown;
    }
    size_t at = b->arenaTop;
    b->arenaTop += count;
    return at;
}

static int rt_nodes_reserve(Octree* tree, uint32_t extra) {
    if ((uint64_t)tree->nodeCount + extra <= tree->nodeCapacity) return 1;
    uint64_t grown = tree->nodeCapacity ? (uint64_t)tree->nodeCapacity * 2 : 64;
    while (grown < (uint64_t)tree->nodeCount + extra) grown *= 2;
    if (grown > 0x7FFFFFFFu) return 0;
    OctreeNode* moved = (OctreeNode*)realloc(tree->nodes, sizeof(OctreeNode) * (size_t)grown);
    if (!moved) return 0;
    tree->nodes = moved;
    tree->nodeCapacity = (uint32_t)grown;
    return 1;
}

static int rt_indexes_reserve(Octree* tree, uint32_t extra) {
    if ((uint64_t)tree->indexCount + extra <= tree->indexCapacity) return 1;
    uint64_t grown = tree->indexCapacity ? (uint64_t)tree->indexCapacity * 2 : 2048;
    while (grown < (uint64_t)tree->indexCount + extra) grown *= 2;
    if (grown > 0xFFFFFFFFu) return 0;
    uint32_t* moved = (uint32_t*)realloc(tree->indexes, sizeof(uint32_t) * (size_t)grown);
    if (!moved) return 0;
    tree->indexes = moved;
    tree->indexCapacity = (uint32_t)grown;
    return 1;
}

/* The eight child boxes in the reference's order: front-bottom-left, front-bottom-right,
 * back-bottom-left, back-bottom-right, front-top-left, front-top-right, back-top-left,
 * back-top-right ("front" = low z, "left" = low x).  Child 6 does not raise its minimum
 * corner in y, so it spans the full height of the parent (reference: src/octree.c:290-345). */
static void rt_child_boxes(const BoundingBox* box, BoundingBox out[8]) {
    Vec3 lo = box->bottomLeftFrontCorner;
    Vec3 half = v3_scale(v3_sub(box->topRightBackCorner, lo), 0.5f);
    Vec3 mid = v3_add(lo, half);
    static const unsigned char shiftX[8] = {0, 1, 0, 1, 0, 1, 0, 1};
    static const unsigned char shiftY[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    static const unsigned char shiftZ[8] = {0, 0, 1, 1, 0, 0, 1, 1};
    for (int c = 0; c < 8; c++) {
        Vec3 d = v3(shiftX[c] ? half.x : 0.0f, shiftY[c] ? half.y : 0.0f, shiftZ[c] ? half.z : 0.0f);
        out[c].bottomLeftFrontCorner = v3_add(lo, d);
        out[c].topRightBackCorner = v3_add(mid, d);
    }
    out[0].bottomLeftFrontCorner = lo;
    out[0].topRightBackCorner = mid;
    out[6].bottomLeftFrontCorner = v3_add(lo, v3(0.0f, 0.0f, half.z));
    out[7].bottomLeftFrontCorner = mid;
    out[7].topRightBackCorner = box->topRightBackCorner;
}

/* sphereAt/triangleAt are arena offsets of the parent's lists. */
static void rt_build_node(Builder* b, uint32_t nodeId, size_t sphereAt, uint32_t sphereIn,
                          size_t triangleAt, uint32_t triangleIn, BoundingBox box) {
    if (b->failed) return;
    Octree* tree = b->tree;
    const Scene* scene = b->scene;
    size_t mark = b->arenaTop;

    size_t keptSpheresAt = rt_arena_reserve(b, sphereIn);
    size_t keptTrianglesAt = rt_arena_reserve(b, triangleIn);
    if (b->failed) return;
    uint32_t keptSpheres = 0, keptTriangles = 0;
    for (uint32_t i = 0; i < sphereIn; i++) {
        uint32_t id = b->arena[sphereAt + i];
        if (rt_sphere_touches_box(&scene->spheres[id], &box)) b->arena[keptSpheresAt + keptSpheres++] = id;
    }
    for (uint32_t i = 0; i < triangleIn; i++) {
        uint32_t id = b->arena[triangleAt + i];
        if (rt_triangle_touches_box(&scene->triangles[id], &box)) b->arena[keptTrianglesAt + keptTriangles++] = id;
    }

    int changed = sphereIn != keptSpheres || triangleIn != keptTriangles || (nodeId == 0 && !b->subtree);
    int small = keptSpheres < MIN_ELEMENTS_PER_NODE && keptTriangles < MIN_ELEMENTS_PER_NODE;

    tree->nodes[nodeId].boundingBox = box;
    if (changed && !small) {
        if (!rt_nodes_reserve(tree, 8)) { b->failed = 1; return; }
        uint32_t firstChild = tree->nodeCount;
        tree->nodeCount += 8;
        OctreeNode* node = &tree->nodes[nodeId];
        node->sphereIndexOffset = node->sphereIndexCount = 0;
        node->triangleIndexOffset = node->triangleIndexCount = 0;
        for (int c = 0; c < 8; c++) node->childNodeIndexes[c] = (int32_t)(firstChild + (uint32_t)c);
        BoundingBox childBox[8];
        rt_child_boxes(&box, childBox);
        for (int c = 0; c < 8; c++) {
            rt_build_node(b, firstChild + (uint32_t)c, keptSpheresAt, keptSpheres,
                          keptTrianglesAt, keptTriangles, childBox[c]);
        }
    } else {
        if (!rt_indexes_reserve(tree, keptSpheres + keptTriangles)) { b->failed = 1; return; }
        OctreeNode* node = &tree->nodes[nodeId];
        for (int c = 0; c < 8; c++) node->childNodeIndexes[c] = NODE_INDEX_UNDEF;
        node->sphereIndexOffset = tree->indexCount;
        node->sphereIndexCount = keptSpheres;
        memcpy(tree->indexes + tree->indexCount, b->arena + keptSpheresAt, sizeof(uint32_t) * keptSpheres);
        tree->indexCount += keptSpheres;
        node->triangleIndexOffset = tree->indexCount;
        node->triangleIndexCount = keptTriangles;
        memcpy(tree->indexes + tree->indexCount, b->arena + keptTrianglesAt, sizeof(uint32_t) * keptTriangles);
        tree->indexCount += keptTriangles;
    }
    b->arenaTop = mark;
}

/* ---- building on several threads ---------------------------------------------------------
 * The sequential builder numbers nodes and appends leaf runs in depth-first order, so the whole
 * subtree under a node occupies one contiguous range of node ids (after the node's own id) and one
 * contiguous range of indexes[].  That makes subtrees independent: the top RT_TOP_LEVELS levels are
 * classified here in order (a few hundred nodes), every node below them becomes a task that builds
 * its subtree into arrays of its own with the same code, and a replay of the depth-first order
 * then gives every top node and every task the ids and offsets the sequential build would have
 * used.  The result is byte-identical for any number of threads. */
#define RT_TOP_LEVELS 3

enum { TOP_LEAF = 0, TOP_INNER = 1, TOP_TASK = 2 };

typedef struct {
    BoundingBox box;
    int kind;
    uint32_t child[8];                 /* TOP_INNER: plan indices of the children */
    uint32_t* spheres;                 /* lists this node kept (owned); a task reads its parent's */
    uint32_t* triangles;
    uint32_t sphereCount, triangleCount;
    uint32_t parent;
    uint32_t finalId, indexOffset;     /* filled by the replay */
    Octree local;                      /* TOP_TASK: the subtree, local numbering (0 = this node) */
    uint32_t nodeBase, indexBase;      /* TOP_TASK: where local node 1 and local index 0 land */
} TopNode;

typedef struct {
    const Scene* scene;
    TopNode* node;
    uint32_t count, capacity;
    int failed;
    volatile uint32_t next;            /* task cursor (atomic) */
    uint32_t* taskList;
    uint32_t taskCount;
    Octree* out;
    int phase;
} Plan;

static uint32_t plan_new_nodes(Plan* plan, uint32_t howMany) {
    if (plan->count + howMany > plan->capacity) {
        uint32_t grown = plan->capacity ? plan->capacity * 2 : 1024;
        while (grown < plan->count + howMany) grown *= 2;
        TopNode* moved = (TopNode*)realloc(plan->node, sizeof(TopNode) * grown);
        if (!moved) { plan->failed = 1; return 0; }
        plan->node = moved;
        plan->capacity = grown;
    }
    uint32_t first = plan->count;
    memset(&plan->node[first], 0, sizeof(TopNode) * howMany);
    plan->count += howMany;
    return first;
}

/* Same decisions as rt_build_node, for the levels above the tasks. */
static void plan_node(Plan* plan, uint32_t self, const uint32_t* spheresIn, uint32_t sphereIn,
                      const uint32_t* trianglesIn, uint32_t triangleIn, BoundingBox box, uint32_t level) {
    if (plan->failed) return;
    const Scene* scene = plan->scene;
    plan->node[self].box = box;
    if (level == RT_TOP_LEVELS) {
        plan->node[self].kind = TOP_TASK;      /* classifies its parent's lists itself, like any node */
        return;
    }
    uint32_t* keptS = (uint32_t*)malloc(sizeof(uint32_t) * (sphereIn + 1u));
    uint32_t* keptT = (uint32_t*)malloc(sizeof(uint32_t) * (triangleIn + 1u));
    if (!keptS || !keptT) { free(keptS); free(keptT); plan->failed = 1; return; }
    uint32_t nS = 0, nT = 0;
    for (uint32_t i = 0; i < sphereIn; i++)
        if (rt_sphere_touches_box(&scene->spheres[spheresIn[i]], &box)) keptS[nS++] = spheresIn[i];
    for (uint32_t i = 0; i < triangleIn; i++)
        if (rt_triangle_touches_box(&scene->triangles[trianglesIn[i]], &box)) keptT[nT++] = trianglesIn[i];
    plan->node[self].spheres = keptS;
    plan->node[self].triangles = keptT;
    plan->node[self].sphereCount = nS;
    plan->node[self].triangleCount = nT;
    int changed = sphereIn != nS || triangleIn != nT || self == 0;
    int small = nS < MIN_ELEMENTS_PER_NODE && nT < MIN_ELEMENTS_PER_NODE;
    if (!(changed && !small)) { plan->node[self].kind = TOP_LEAF; return; }
    uint32_t first = plan_new_nodes(plan, 8);
    if (plan->failed) return;
    plan->node[self].kind = TOP_INNER;
    BoundingBox childBox[8];
    rt_child_boxes(&box, childBox);
    for (int c = 0; c < 8; c++) {
        plan->node[self].child[c] = first + (uint32_t)c;
        plan->node[first + (uint32_t)c].parent = self;
    }
    for (int c = 0; c < 8; c++)
        plan_node(plan, first + (uint32_t)c, keptS, nS, keptT, nT, childBox[c], level + 1);
}

static void task_build(Plan* plan, TopNode* t) {
    const TopNode* parent = &plan->node[t->parent];
    Builder b;
    memset(&b, 0, sizeof b);
    b.tree = &t->local;
    b.scene = plan->scene;
    b.subtree = 1;
    b.arenaCapacity = (size_t)parent->sphereCount + parent->triangleCount + 1024;
    b.arena = (uint32_t*)malloc(b.arenaCapacity * sizeof(uint32_t));
    if (!b.arena || !rt_nodes_reserve(b.tree, 1) || !rt_indexes_reserve(b.tree, 1)) { free(b.arena); plan->failed = 1; return; }
    size_t spheresAt = rt_arena_reserve(&b, parent->sphereCount);
    size_t trianglesAt = rt_arena_reserve(&b, parent->triangleCount);
    memcpy(b.arena + spheresAt, parent->spheres, sizeof(uint32_t) * parent->sphereCount);
    memcpy(b.arena + trianglesAt, parent->triangles, sizeof(uint32_t) * parent->triangleCount);
    b.tree->nodeCount = 1;
    rt_build_node(&b, 0, spheresAt, parent->sphereCount, trianglesAt, parent->triangleCount, t->box);
    free(b.arena);
    if (b.failed) plan->failed = 1;
}

/* Copies a task's subtree to its place in the final arrays, translating ids and offsets. */
static void task_place(Plan* plan, TopNode* t) {
    Octree* out = plan->out;
    const Octree* local = &t->local;
    for (uint32_t j = 0; j < local->nodeCount; j++) {
        OctreeNode n = local->nodes[j];
        if (n.childNodeIndexes[0] != NODE_INDEX_UNDEF) {
            for (int c = 0; c < 8; c++) n.childNodeIndexes[c] = (int32_t)(t->nodeBase + (uint32_t)n.childNodeIndexes[c] - 1u);
        } else {
            n.sphereIndexOffset += t->indexBase;
            n.triangleIndexOffset += t->indexBase;
        }
        out->nodes[j == 0 ? t->finalId : t->nodeBase + j - 1u] = n;
    }
    if (local->indexCount) memcpy(out->indexes + t->indexBase, local->indexes, sizeof(uint32_t) * local->indexCount);
}

static void* plan_worker(void* arg) {
    Plan* plan = (Plan*)arg;
    for (;;) {
        uint32_t i = __atomic_fetch_add(&plan->next, 1u, __ATOMIC_RELAXED);
        if (i >= plan->taskCount) return NULL;
        TopNode* t = &plan->node[plan->taskList[i]];
        if (plan->phase == 0) task_build(plan, t);
        else task_place(plan, t);
    }
}

static void plan_run(Plan* plan, int phase, unsigned threads) {
    plan->phase = phase;
    plan->next = 0;
    pthread_t pool[64];
    unsigned started = 0;
    for (unsigned i = 1; i < threads && i < 64; i++)
        if (pthread_create(&pool[started], NULL, plan_worker, plan) == 0) started++;
    plan_worker(plan);
    for (unsigned i = 0; i < started; i++) pthread_join(pool[i], NULL);
}

/* Depth-first replay: hands out node ids and index offsets in the sequential builder's order. */
static void plan_replay(Plan* plan, uint32_t self, uint32_t* nodeCursor, uint32_t* indexCursor) {
    TopNode* t = &plan->node[self];
    if (t->kind == TOP_INNER) {
        uint32_t first = *nodeCursor;
        *nodeCursor += 8;
        for (int c = 0; c < 8; c++) plan->node[t->child[c]].finalId = first + (uint32_t)c;
        for (int c = 0; c < 8; c++) plan_replay(plan, plan->node[self].child[c], nodeCursor, indexCursor);
    } else if (t->kind == TOP_LEAF) {
        t->indexOffset = *indexCursor;
        *indexCursor += t->sphereCount + t->triangleCount;
    } else {
        t->nodeBase = *nodeCursor;
        *nodeCursor += t->local.nodeCount - 1u;
        t->indexBase = *indexCursor;
        *indexCursor += t->local.indexCount;
    }
}

static unsigned build_threads(const Scene* scene) {
    long online = sysconf(_SC_NPROCESSORS_ONLN);
    unsigned threads = online > 16 ? 16u : online < 1 ? 1u : (unsigned)online;
    const char* env = getenv("RT_BUILD_THREADS");
    if (env && atoi(env) > 0) threads = (unsigned)atoi(env);
    else if ((uint64_t)scene->sphereCount + scene->triangleCount < 20000) threads = 1;
    return threads > 64 ? 64 : threads;
}

Octree* rt_octree_buildFromScene(Scene* scene) {
    if (!scene) return NULL;
    if ((scene->sphereCount && !scene->spheres) || (scene->triangleCount && !scene->triangles)) return NULL;
    Plan plan;
    memset(&plan, 0, sizeof plan);
    plan.scene = scene;
    Octree* tree = NULL;
    uint32_t* allSpheres = (uint32_t*)malloc(sizeof(uint32_t) * (scene->sphereCount + 1u));
    uint32_t* allTriangles = (uint32_t*)malloc(sizeof(uint32_t) * (scene->triangleCount + 1u));
    if (!allSpheres || !allTriangles) plan.failed = 1;
    if (!plan.failed) {
        for (uint32_t i = 0; i < scene->sphereCount; i++) allSpheres[i] = i;
        for (uint32_t i = 0; i < scene->triangleCount; i++) allTriangles[i] = i;
        plan_new_nodes(&plan, 1);
    }
    if (!plan.failed)
        plan_node(&plan, 0, allSpheres, scene->sphereCount, allTriangles, scene->triangleCount, rt_root_box(scene), 0);
    if (!plan.failed) {
        plan.taskList = (uint32_t*)malloc(sizeof(uint32_t) * (plan.count + 1u));
        if (!plan.taskList) plan.failed = 1;
    }
    const unsigned threads = build_threads(scene);
    if (!plan.failed) {
        for (uint32_t i = 0; i < plan.count; i++)
            if (plan.node[i].kind == TOP_TASK) plan.taskList[plan.taskCount++] = i;
        plan_run(&plan, 0, threads);
    }
    if (!plan.failed) {
        uint32_t nodeCursor = 1, indexCursor = 0;
        plan.node[0].finalId = 0;
        plan_replay(&plan, 0, &nodeCursor, &indexCursor);
        tree = (Octree*)calloc(1, sizeof(Octree));
        if (tree) {
            tree->nodes = (OctreeNode*)malloc(sizeof(OctreeNode) * nodeCursor);
            tree->indexes = (uint32_t*)malloc(sizeof(uint32_t) * (indexCursor ? indexCursor : 1u));
            tree->nodeCount = tree->nodeCapacity = nodeCursor;
            tree->indexCount = indexCursor;
            tree->indexCapacity = indexCursor ? indexCursor : 1u;
        }
        if (!tree || !tree->nodes || !tree->indexes) {
            rt_octree_destroy(tree);
            tree = NULL;
        } else {
            for (uint32_t i = 0; i < plan.count; i++) {
                const TopNode* t = &plan.node[i];
                if (t->kind == TOP_TASK) continue;
                OctreeNode* node = &tree->nodes[t->finalId];
                memset(node, 0, sizeof *node);
                node->boundingBox = t->box;
                if (t->kind == TOP_INNER) {
                    for (int c = 0; c < 8; c++) node->childNodeIndexes[c] = (int32_t)plan.node[t->child[c]].finalId;
                } else {
                    for (int c = 0; c < 8; c++) node->childNodeIndexes[c] = NODE_INDEX_UNDEF;
                    node->sphereIndexOffset = t->indexOffset;
                    node->sphereIndexCount = t->sphereCount;
                    node->triangleIndexOffset = t->indexOffset + t->sphereCount;
                    node->triangleIndexCount = t->triangleCount;
                    memcpy(tree->indexes + t->indexOffset, t->spheres, sizeof(uint32_t) * t->sphereCount);
                    memcpy(tree->indexes + t->indexOffset + t->sphereCount, t->triangles, sizeof(uint32_t) * t->triangleCount);
                }
            }
            plan.out = tree;
            plan_run(&plan, 1, threads);
        }
    }
    for (uint32_t i = 0; i < plan.count; i++) {
        free(plan.node[i].spheres);
        free(plan.node[i].triangles);
        free(plan.node[i].local.nodes);
        free(plan.node[i].local.indexes);
    }
    free(plan.node);
    free(plan.taskList);
    free(allSpheres);
    free(allTriangles);
    return tree;
}

void rt_octree_destroy(Octree* octree) {
    if (!octree) return;
    free(octree->nodes);
    free(octree->indexes);
    free(octree);
}

uint32_t rt_octree_levels(const Octree* octree) {
    if (!octree || !octree->nodes || octree->nodeCount == 0) return 0;
    /* Children always have larger ids than their parent, so one forward pass suffices. */
    uint32_t* level = (uint32_t*)calloc(octree->nodeCount, sizeof(uint32_t));
    if (!level) return 0;
    uint32_t deepest = 0;
    for (uint32_t n = 0; n < octree->nodeCount; n++) {
        const OctreeNode* node = &octree->nodes[n];
        if (level[n] > deepest) deepest = level[n];
        if (node->childNodeIndexes[0] == NODE_INDEX_UNDEF) continue;
        for (int c = 0; c < 8; c++) {
            int32_t child = node->childNodeIndexes[c];
            if (child <= (int32_t)n || (uint32_t)child >= octree->nodeCount) { free(level); return 0; }
            level[child] = level[n] + 1;
        }
    }
    free(level);
    return deepest + 1;
}
