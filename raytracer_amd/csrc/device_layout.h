/*
 * device_layout.h -- the private device copy of a scene, shared by the flattening
 * code (host) and the kernels (device).
 *
 * The device walks "wide nodes": one record per INNER node holding what a ray needs to
 * choose among up to eight children without touching them.  They are produced either from the
 * scene's primitives alone or from the reference's octree (reference: src/octree.h:15-34:
 * 72-byte nodes with eight child ids, leaves pointing at runs of indexes[]; children are always
 * eight consecutive ids and a walk always considers all eight, src/octree.c:270-286,
 * src/kernel.cl:566-569, so the per-child records collapse into struct-of-arrays rows).
 * WideNode (float boxes) is the host-side intermediate; QNode is what is uploaded.
 */
#ifndef RT_DEVICE_LAYOUT_H
#define RT_DEVICE_LAYOUT_H

#include <stdint.h>

/* One inner node.  Boxes are TIGHT (bounds of the primitives that reach the child,
 * clipped to the child's cell) and padded outwards, not the reference's cells; a child
 * with nothing in it has lo > hi on every axis and can never be entered.  Rows are
 * indexed [axis][child]; child order is the reference's (bit0 = +x, bit1 = +z, bit2 = +y). */
struct WideNode {
    float lo[3][8];     /*   0: minimum corner per child */
    float hi[3][8];     /*  96: maximum corner per child */
    struct Ref {
        uint32_t link;  /* inner child: index of its WideNode; leaf: offset of its run in the
                           leaf arrays (spheres first, then triangles)                       */
        uint32_t kind;  /* WIDE_INNER, or for a leaf (sphereCount << 20) | triangleCount       */
    } ref[8];           /* 192: one 8-byte load per child visited                            */
};
static_assert(sizeof(WideNode) == 256, "WideNode is two 128-byte cache lines");

#define WIDE_INNER 0x80000000u
#define WIDE_MAX_LEAF_TRIANGLES ((1u << 20) - 1u)
#define WIDE_MAX_LEAF_SPHERES ((1u << 11) - 1u)

/* The form of a WideNode the kernels read: ONE 128-byte cache line.  The eight child boxes are
 * quantised to 8 bits per coordinate on a per-node grid (origin + scale * q), minimum corners
 * rounded down and maximum corners up, so a quantised box always contains the float box it
 * came from (which already contains the primitives plus padding): culling stays conservative
 * and loses at most 1/255 of the node's extent per side.  Five loads fetch everything a node
 * test needs (48 B of boxes + 24 B of grid) instead of twelve, and the entry/exit planes of a
 * ray are chosen by selecting packed words, not by per-ray row addresses.  The fifth load
 * (bytes 64-79) also brings firstInner and innerMask.
 * An empty child has q = 255 for its minimum and 0 for its maximum on every axis. */
struct QNode {
    uint8_t qlo[3][8];   /*   0: [axis][child] */
    uint8_t qhi[3][8];   /*  24 */
    float origin[3];     /*  48 */
    float scale[3];      /*  60 */
    uint32_t firstInner; /*  72: index of the node of the first inner child; the inner children of a node
                                 are consecutive nodes in child order, so inner child c is node
                                 firstInner + popcount(innerMask & ((1 << c) - 1)): a walk descends
                                 without reading link[] or kind[] */
    uint32_t innerMask;  /*  76: bit c set = child c is an inner node (bits 8-31 zero) */
    uint32_t link[8];    /*  80: as WideNode::Ref::link (read for leaf children) */
    uint16_t kind[8];    /* 112: QNODE_INNER, or (sphereCount << 11) | triangleCount */
};
static_assert(sizeof(QNode) == 128, "QNode is one 128-byte cache line");
#define QNODE_INNER 0x8000u
#define QNODE_MAX_LEAF_TRIANGLES 2047u
#define QNODE_MAX_LEAF_SPHERES 15u

struct Float4 { float x, y, z, w; };

/* One entry of a leaf's triangle run: a 64-byte FILTER record stored in leaf order, so that a
 * leaf of four triangles is 256 contiguous bytes and rejecting a triangle needs no dependent
 * index -> triangle gather.  A triangle that reaches several leaves is stored once per leaf;
 * HBM capacity is spent to remove a load dependency.
 *   plane = (n.xyz, n . v0): the unit normal and offset computed at upload with the reference's
 *           own expression (reference: src/raytracer.c:119-121,127), bit-identical to a per-test
 *           recomputation, so the exact test reuses them;
 *   edge[k] = (m_k.xyz, c_k) with m_k = (n x e_k) / |e_k| and c_k = m_k . v_k for the reference's
 *           three edges (e_0 = v1-v0 at v0, e_1 = v2-v1 at v1, e_2 = v0-v2 at v2): m_k . P - c_k
 *           is the signed in-plane distance of P from edge k, >= 0 inside.  Computed in fp64 and
 *           rounded; used only to REJECT with a margin, never to accept. */
struct LeafTriangle {
    Float4 plane;
    Float4 edge[3];
};
static_assert(sizeof(LeafTriangle) == 64, "LeafTriangle is half a cache line");

/* Per-light direction grids (light_grid.hip).  Light l has 6 R^2 buckets, bucket g = face * R^2 + w * R + u, and a
 * stretch of refs beginning at its first reference F(l).  Eight consecutive buckets share one 12-byte block of the
 * light's index: the offset of the first one's list in the light's stretch and the eight lists' lengths, one byte
 * each (a light with a list of more than 255 entries has no grid); bucket g lists
 * refs[F(l) + base + lengths before g in its block ...), length[g & 7] entries.  An entry: in its low 24 bits a sphere's
 * number (GRID_REF_SPHERE set) or a triangle's index into leafTriangle -- its position in the leaf arrays where every
 * triangle is in one leaf (the library's own hierarchy), its own number where the records are per triangle (the
 * caller's octree); in its top seven bits the primitive's distance from the light, rounded down to one of 128 levels
 * (level = floor(lower bound of the distance * the light's levels per unit length)).  A list is sorted by the whole
 * word, nearest level first: a ray from a point at distance D from the light is done with its list at the first entry
 * whose level lies beyond D -- nothing behind the shading point, seen from the light, is ever tested. */
#define GRID_REF_INDEX 0x00FFFFFFu
#define GRID_REF_SPHERE 0x01000000u
#define GRID_REF_LEVEL_SHIFT 25
#define GRID_REF_LEVELS 128u
#define GRID_REF_MAX_ENTRIES (1u << 24)   /* scenes with more spheres + triangles than this get no grids */
struct LightGrid {
    const uint32_t* blocks;     /* [light * blocksPerLight + g / 8] x 3 words: base, lengths of buckets 0-3, of buckets 4-7 */
    const uint32_t* refs;
    const uint32_t* always;     /* [light * RT_GRID_ALWAYS + i]: primitives tested for every ray towards the light */
    const uint32_t* info;       /* RT_GRID_INFO_WORDS per light: has a grid (0/1), entries of its always list, F(l) low and high word,
                                   levels per unit length (a float's bits) */
    uint32_t resolution;        /* R */
    float halfResolution;       /* R / 2, as a float (a wave-uniform value the kernel would otherwise convert in a vector register) */
    uint32_t blocksPerLight;    /* (6 R^2 + 7) / 8 */
    uint32_t allLights;         /* non-zero: every light of the scene has a grid */
    float reach;                /* a ray that starts at a larger |coordinate| than this walks the hierarchy instead */
};

struct DeviceScene {
    const QNode* nodes;
    const WideNode* wideNodes;     /* the same nodes with float boxes, same numbering: read by the wave-shared walk of
                                      coherent rays through scalar loads (one node for the whole wavefront); NULL = none */
    const uint32_t* leafIndex;     /* the reference's Octree.indexes[] + runs of refined leaves */
    const LeafTriangle* leafTriangle; /* same indexing as leafIndex; filled for triangle entries */
    const Float4* trianglePlane;   /* per triangle: (n.xyz, n . v0), read once per hit for the normal */
    const Float4* triangleCorner;  /* per triangle: 3 x (v.xyz, -), read by the exact test only */
    const uint32_t* triangleMaterial;
    const Float4* sphere;          /* (centre.xyz, radius) */
    const uint32_t* sphereMaterial;
    const Float4* plane;           /* (normal.xyz, distanceFromOrigin) */
    const uint32_t* planeMaterial;
    const Float4* material;        /* 3 per material: (reflection, refraction, ambient weight, diffuse
                                      weight), (color.rgb, specular exponent), (specular weight, -, -, -) */
    const Float4* light;           /* 2 per light: (position.xyz, strength), (emission.rgb, -) */
    uint32_t planeCount, sphereCount, triangleCount, lightCount, materialCount;
    uint32_t treeLevels;           /* inner levels a walk can be inside at once */
    float filterMargin;            /* absolute slack of the conservative (fused-arithmetic) tests: covers the
                                      rounding of a dot product whose terms are as large as the scene -- or, per
                                      frame, as large as the camera's distance from the origin */
    LightGrid grid;                /* start == NULL: none */
    float boxPad;                  /* per frame: extra widening of every box for rays that start far outside the
                                      scene (0 when the padding applied at upload suffices) */
};

#endif
