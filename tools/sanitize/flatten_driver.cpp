#include "flatten.h"
#include "rt_host.h"
#include <cstdio>
int main() {
    struct { uint32_t qx, qz, lights, spheres; } cases[] = {{40, 25, 3, 12}, {150, 120, 8, 500}, {2, 2, 0, 0}, {90, 60, 70, 30}};
    for (auto& c : cases) {
        Scene* s = rt_scene_synth_mesh(64, 36, c.qx, c.qz, 0, c.lights, c.spheres);
        Octree* t = rt_octree_buildFromScene(s);
        for (uint32_t accel = 0; accel < 2; accel++) {
            FlatScene flat; std::string err;
            int rc = rt_flatten(s, accel ? t : nullptr, accel, flat, err);
            HipTraceHierarchyReport rep;
            if (rc == 0) rt_check_hierarchy(s, flat, accel, rep);
            std::printf("%ux%u accel %u rc %d nodes %zu levels %u unreached %u/%u escaped %u bad %u %s\n", c.qx, c.qz, accel, rc, flat.qnodes.size(), flat.treeLevels,
                        rep.unreachedSpheres, rep.unreachedTriangles, rep.escapedPrimitives, rep.badLinks, err.c_str());
        }
        rt_octree_destroy(t); rt_scene_destroy(s);
    }
    Scene* e = rt_scene_create(); FlatScene flat; std::string err;
    std::printf("empty rc %d\n", rt_flatten(e, nullptr, 0, flat, err));
    rt_scene_destroy(e);
    return 0;
}
