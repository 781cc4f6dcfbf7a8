// host_capi.cpp -- C entry points of libraytracer_host.so for non-C++ hosts (the Python package, tests, bench).
// Thin: builds one of the built-in scenes through the scene-definition surface (Scenes.cpp), flattens it
// (Scene.cpp) and either hands the flat arrays out (for the parity tests, which feed the SAME arrays to the
// oracle and to the C ABI) or uploads them to a rt_context.
#include <cstring>
#include <string>
#include "Scene.h"

namespace {
thread_local std::string g_err;
struct HostScene {
    FlatScene flat;
    Camera camera;
    std::string textureDir;
};
template <class T>
void copyOut(T *dst, const std::vector<T> &src) {
    if (dst && !src.empty()) std::memcpy(dst, src.data(), src.size() * sizeof(T));
}
}  // namespace

extern "C" {

const char *rth_last_error(void) { return g_err.c_str(); }

// name: "sample" | "bunny" | "competition" | "cones" | "stress:<replicas>" | "stress-diagonal:<replicas>" (second camera); asset_dir holds meshes/ and textures/
void *rth_scene_create(const char *name, const char *asset_dir) {
    try {
        clearScene();
        HostScene *hs = new HostScene();
        const std::string n = name, dir = asset_dir, meshes = dir + "/meshes";
        if (n == "sample") hs->camera = sampleScene(meshes);
        else if (n == "bunny") hs->camera = bunnyScene(meshes);
        else if (n == "competition") hs->camera = competitionScene(meshes);
        else if (n == "cones") hs->camera = coneScene(meshes);
        else if (n.rfind("stress:", 0) == 0) hs->camera = stressScene(meshes, std::atoi(n.c_str() + 7));
        else if (n.rfind("stress-diagonal:", 0) == 0) hs->camera = stressScene(meshes, std::atoi(n.c_str() + 16), true);
        else throw std::runtime_error("unknown scene '" + n + "'");
        hs->flat = flattenScene(objects, lights, ambient_light);
        hs->textureDir = dir + "/textures";
        clearScene();
        return hs;
    } catch (std::exception &e) {
        g_err = e.what();
        clearScene();
        return nullptr;
    }
}

void rth_scene_destroy(void *h) { delete (HostScene *)h; }

// counts[8] = triangles, spheres, planes, materials, objects, lights, uses_bmp, cones
int rth_scene_counts(void *h, int64_t *counts) {
    const FlatScene &s = ((HostScene *)h)->flat;
    counts[0] = s.triangleCount();
    counts[1] = (int64_t)s.sph_object.size();
    counts[2] = (int64_t)s.pl_object.size();
    counts[3] = (int64_t)s.materials.size();
    counts[4] = (int64_t)s.object_material.size();
    counts[5] = (int64_t)s.light_pos.size() / 3;
    counts[6] = s.uses_bmp ? 1 : 0;
    counts[7] = (int64_t)s.cone_object.size();
    return 0;
}

int rth_scene_camera(void *h, rt_camera *out) {
    *out = ((HostScene *)h)->camera.toC();
    return 0;
}

int rth_scene_get_triangles(void *h, float *v0, float *v1, float *v2, float *nA, float *nB, float *nC, float *tA,
                            float *tB, float *tC, uint32_t *flags, int32_t *object_id) {
    const FlatScene &s = ((HostScene *)h)->flat;
    copyOut(v0, s.v0); copyOut(v1, s.v1); copyOut(v2, s.v2);
    copyOut(nA, s.nA); copyOut(nB, s.nB); copyOut(nC, s.nC);
    copyOut(tA, s.tA); copyOut(tB, s.tB); copyOut(tC, s.tC);
    copyOut(flags, s.tri_flags);
    copyOut(object_id, s.tri_object);
    return 0;
}

int rth_scene_get_analytic(void *h, float *sph_M, float *sph_Minv, float *sph_N, int32_t *sph_object, float *pl_point,
                           float *pl_normal, int32_t *pl_object) {
    const FlatScene &s = ((HostScene *)h)->flat;
    copyOut(sph_M, s.sph_M); copyOut(sph_Minv, s.sph_Minv); copyOut(sph_N, s.sph_N); copyOut(sph_object, s.sph_object);
    copyOut(pl_point, s.pl_point); copyOut(pl_normal, s.pl_normal); copyOut(pl_object, s.pl_object);
    return 0;
}

int rth_scene_get_cones(void *h, float *M, float *Minv, float *Nmat, int32_t *object_id) {
    const FlatScene &s = ((HostScene *)h)->flat;
    copyOut(M, s.cone_M); copyOut(Minv, s.cone_Minv); copyOut(Nmat, s.cone_N); copyOut(object_id, s.cone_object);
    return 0;
}

int rth_scene_get_shading(void *h, rt_material *materials, int32_t *object_material, float *light_pos,
                          float *light_color, float *ambient) {
    const FlatScene &s = ((HostScene *)h)->flat;
    copyOut(materials, s.materials);
    copyOut(object_material, s.object_material);
    copyOut(light_pos, s.light_pos);
    copyOut(light_color, s.light_color);
    if (ambient) std::memcpy(ambient, s.ambient, sizeof s.ambient);
    return 0;
}

// decode one BMP as the renderer expects it (top-down, BGR); query with bytes == NULL first
int rth_read_bmp(const char *path, int32_t *w, int32_t *h, int32_t *channels, uint8_t *bytes) {
    try {
        BmpImage img = readBmp(path);
        *w = img.width; *h = img.height; *channels = img.channels;
        if (bytes) std::memcpy(bytes, img.pixels.data(), img.pixels.size());
        return 0;
    } catch (std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

// parse one .obj with MeshLoader semantics; counts[4] = triangles, v, vn, vt; then fetch with rth_scene_get_*
void *rth_mesh_create(const char *path, float tx, float ty, float tz, int64_t *counts) {
    try {
        MeshLoader ml(path, glm::vec3(tx, ty, tz), false);
        HostScene *hs = new HostScene();
        ml.flatten(hs->flat, 0);
        counts[0] = (int64_t)ml.getTriangles().size();
        counts[1] = (int64_t)ml.vertexCount();
        counts[2] = (int64_t)ml.normalCount();
        counts[3] = (int64_t)ml.uvCount();
        return hs;
    } catch (std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

// flatten -> rt_set_* -> rt_commit on an existing context
// test hook: the .obj scanner's %f on one string: value, characters consumed (0 = no conversion), and whether the
// fast path (not strtof) produced it
int rth_scan_float(const char *text, float *value, int32_t *consumed, int32_t *fast) {
    const char *e = text;
    float v = 0.0f;
    const bool f = objFastFloat(text, &e, &v);
    if (!f) {
        char *e2;
        v = strtof(text, &e2);
        e = e2;
    }
    *value = v;
    *consumed = (int32_t)(e - text);
    *fast = f ? 1 : 0;
    return 0;
}

int rth_scene_upload(void *h, rt_context *ctx) {
    try {
        HostScene *hs = (HostScene *)h;
        uploadScene(ctx, hs->flat, hs->textureDir);
        return 0;
    } catch (std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

}  // extern "C"
