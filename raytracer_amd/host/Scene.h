// Scene.h -- the seam between the scene-definition surface and the GPU renderer.
//
// The reference renders from three mutable globals, `lights`, `ambient_light`, `objects` (main.cpp:36-38),
// inside main()'s tile loop (main.cpp:607-650).  The same globals exist here; `render()` stands where that loop
// stood and `trace_ray()` keeps its signature (main.cpp:221).  Both flatten the globals into SoA buffers
// (FlatScene), hand them to the C ABI (include/rt_hip.h) and run the HIP wavefront pipeline.  Nothing here
// intersects or shades on the CPU.
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>
#include "BmpReader.h"
#include "Image.h"
#include "MeshLoader.h"
#include "Objects.h"

class Light {
public:
    glm::vec3 position;
    glm::vec3 color;
    explicit Light(glm::vec3 p) : position(p), color(glm::vec3(1.0f)) {}
    Light(glm::vec3 p, glm::vec3 c) : position(p), color(c) {}
};

extern std::vector<Light *> lights;
extern glm::vec3 ambient_light;
extern std::vector<Object *> objects;

// Camera of the render loop: main.cpp:570-578 (origin, rotation) and :634/:637 (sign of dz).
struct Camera {
    glm::vec3 origin = glm::vec3(0.0f);
    glm::mat4 rotation = glm::mat4(1.0f);
    float dz = 1.0f;
    float fov = 90.0f;
    rt_camera toC() const {
        rt_camera c{};
        c.origin[0] = origin.x; c.origin[1] = origin.y; c.origin[2] = origin.z;
        c.dz = dz;
        c.fov_deg = fov;
        for (int k = 0; k < 4; k++)
            for (int r = 0; r < 4; r++) c.rot[4 * k + r] = rotation[k][r];
        return c;
    }
};

// SoA image of a scene, exactly the argument lists of rt_set_* (include/rt_hip.h)
struct FlatScene {
    std::vector<rt_material> materials;
    std::vector<int32_t> object_material;
    std::vector<float> v0, v1, v2, nA, nB, nC, tA, tB, tC;
    std::vector<uint32_t> tri_flags;
    std::vector<int32_t> tri_object;
    std::vector<float> sph_M, sph_Minv, sph_N;
    std::vector<int32_t> sph_object;
    std::vector<float> cone_M, cone_Minv, cone_N;
    std::vector<int32_t> cone_object;
    std::vector<float> pl_point, pl_normal;
    std::vector<int32_t> pl_object;
    std::vector<float> light_pos, light_color;
    float ambient[3] = {0, 0, 0};
    bool uses_bmp = false;

    int64_t triangleCount() const { return (int64_t)tri_flags.size(); }
    int addMaterial(const Material &m);  // returns material index (deduplicated)
};

FlatScene flattenScene(const std::vector<Object *> &objs, const std::vector<Light *> &ls, glm::vec3 ambient);

// Uploads a flattened scene (+ the four BMP maps from `texture_dir` when a material needs them) and commits.
void uploadScene(rt_context *ctx, const FlatScene &s, const std::string &texture_dir, bool commit = true);

// Lazily created process-wide context on HIP device `device` (default: RT_DEVICE env or 0).
rt_context *defaultContext(int device = -1);  // device >= 0: that GPU (replaces a context on another one); -1: RT_DEVICE, else 0

struct RenderOptions {
    int spp = 4;        // main.cpp:625
    int maxDepth = 3;   // main.cpp:643
    uint32_t seed = 0;
    int jitterMode = RT_JITTER_AUTO;
    std::string textureDir = "./textures";
    // HIP devices to render on.  Empty or one entry: one GPU (RT_DEVICE or 0).  More: the frame's tiles are interleaved
    // over these GPUs and gathered with RCCL (rt_multi_render) -- what `#pragma omp parallel for` was at main.cpp:607.
    std::vector<int> devices;
    // one image and no more (the reference's main()): the library does not size its queues for a repeat of the frame
    bool singleFrame = false;
};

// Stands where main.cpp:607-650 stood: renders the global scene into `image` on the GPU.
rt_stats render(Image &image, const Camera &camera, const RenderOptions &opt = RenderOptions());

// main.cpp:221 -- colour along one ray, bounce depth `bounces`; runs one ray through the GPU pipeline.
glm::vec3 trace_ray(Ray ray, int bounces);
// Batched form; also returns the closest hits when `hits` is given.
std::vector<glm::vec3> trace_rays(const std::vector<Ray> &rays, int bounces, std::vector<Hit> *hits = nullptr);

// Where the time between "program start" and "first pixel" went (the reference's one timer lumps all of it with the
// render, main.cpp:541,651-653): filled by the built-in scene functions (parse) and by render()'s upload; printed by
// raytracer_hip and under RT_VERBOSE=1.
struct IngestTimings {
    int meshes = 0;
    double ms_parse = 0;      // .obj files -> Triangle lists (MeshLoader ctor; files parsed concurrently)
    double ms_flatten = 0;    // object list -> SoA arrays of the C ABI
    double ms_bmp = 0;        // BMP decode of the four maps (only when a material samples them)
    double ms_set = 0;        // rt_set_*: copies into the context
    rt_commit_stats commit{}; // BVH build / reference order / records / upload (rt_get_commit_stats)
};
IngestTimings &lastIngest();
double ingestNowMs();
void printIngest(std::FILE *f);

// Built-in scenes, restated from the reference through the surface above (Scenes.cpp).
Camera sampleScene(const std::string &meshDir = "./meshes");         // main.cpp:258-407
Camera bunnyScene(const std::string &meshDir = "./meshes");          // sampleScene minus the spheres
Camera competitionScene(const std::string &meshDir = "./meshes");    // main.cpp:409-536 minus the missing pietre.obj
Camera coneScene(const std::string &meshDir = "./meshes");             // Cone test scene (oracle/ref_harness.cpp: cones_scene)
Camera stressScene(const std::string &meshDir, int replicas, bool diagonal = false);        // competition meshes on a lattice
void clearScene();
// render()/trace_ray() re-upload the globals whenever their fingerprint changes (objects, materials, transformations,
// lights, ambient_light, texture directory); invalidateScene() forces it for edits the fingerprint cannot see.
void invalidateScene();
// directory of the four BMP maps (Textures.h:168-171) for trace_ray()/trace_rays(); render() takes RenderOptions::textureDir
void setTextureDir(const std::string &dir);
