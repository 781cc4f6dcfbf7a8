// Scene.cpp -- flattening of the object list and the host side of render()/trace_ray().  See Scene.h.
#include "Scene.h"
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>

std::vector<Light *> lights;
glm::vec3 ambient_light(0.7f);  // main.cpp:37
std::vector<Object *> objects;

IngestTimings &lastIngest() {
    static IngestTimings t;
    return t;
}
double ingestNowMs() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void rtCheck(int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + rt_last_error());
}

// ---- texture tags -----------------------------------------------------------------------------------
static rt_context *g_tex_ctx = nullptr;
void rt_set_host_texture_context(rt_context *ctx) { g_tex_ctx = ctx; }
static glm::vec3 evalOnGpu(int id, glm::vec2 uv) {
    rt_context *ctx = g_tex_ctx ? g_tex_ctx : defaultContext();
    float in[2] = {uv.x, uv.y}, out[3];
    rtCheck(rt_eval_texture(ctx, id, 1, in, out), "rt_eval_texture");
    return glm::vec3(out[0], out[1], out[2]);
}
glm::vec3 perlinNormal(glm::vec2 uv) { return evalOnGpu(RT_TEX_PERLIN_NORMAL, uv); }
glm::vec3 perlinWater(glm::vec2 uv) { return evalOnGpu(RT_TEX_PERLIN_WATER, uv); }
glm::vec3 opal(glm::vec2 uv) { return evalOnGpu(RT_TEX_OPAL, uv); }
glm::vec3 perlinTerrain(glm::vec2 uv) { return evalOnGpu(RT_TEX_PERLIN_TERRAIN, uv); }
glm::vec3 perlinIceTerrain(glm::vec2 uv) { return evalOnGpu(RT_TEX_PERLIN_ICE_TERRAIN, uv); }
glm::vec3 snowTerrain(glm::vec2 uv) { return evalOnGpu(RT_TEX_SNOW_TERRAIN, uv); }
glm::vec3 qwilfishTexture(glm::vec2 uv) { return evalOnGpu(RT_TEX_QWILFISH, uv); }
glm::vec3 colorAt(glm::vec2 uv) { return evalOnGpu(RT_TEX_BMP_COLOR, uv); }
glm::vec3 normalAt(glm::vec2 uv) { return evalOnGpu(RT_TEX_BMP_NORMAL, uv); }
float ambientOcclusionAt(glm::vec2 uv) { return evalOnGpu(RT_TEX_BMP_OCCLUSION, uv).x; }
float roughnessAt(glm::vec2 uv) { return evalOnGpu(RT_TEX_BMP_ROUGHNESS, uv).x; }

int rt_texture_id_of(glm::vec3 (*f)(glm::vec2)) {
    if (f == nullptr) return RT_TEX_NONE;
    if (f == &perlinNormal) return RT_TEX_PERLIN_NORMAL;
    if (f == &perlinWater) return RT_TEX_PERLIN_WATER;
    if (f == &opal) return RT_TEX_OPAL;
    if (f == &perlinTerrain) return RT_TEX_PERLIN_TERRAIN;
    if (f == &perlinIceTerrain) return RT_TEX_PERLIN_ICE_TERRAIN;
    if (f == &snowTerrain) return RT_TEX_SNOW_TERRAIN;
    if (f == &qwilfishTexture) return RT_TEX_QWILFISH;
    if (f == &colorAt) return RT_TEX_BMP_COLOR;
    if (f == &normalAt) return RT_TEX_BMP_NORMAL;
    return -1;
}
int rt_texture_id_of(float (*f)(glm::vec2)) {
    if (f == nullptr) return RT_TEX_NONE;
    if (f == &roughnessAt) return RT_TEX_BMP_ROUGHNESS;
    if (f == &ambientOcclusionAt) return RT_TEX_BMP_OCCLUSION;
    return -1;
}

// ---- flattening -------------------------------------------------------------------------------------
int FlatScene::addMaterial(const Material &m) {
    rt_material r;
    std::memset(&r, 0, sizeof r);
    for (int k = 0; k < 3; k++) {
        r.ambient[k] = m.ambient[k];
        r.diffuse[k] = m.diffuse[k];
        r.specular[k] = m.specular[k];
    }
    r.reflection = m.reflection;
    r.refraction = m.refraction;
    r.sigma = m.sigma;
    r.shininess = m.shininess;
    r.texture = rt_texture_id_of(m.texture);
    r.has_normal_map = m.hasNormalMap ? 1 : 0;
    r.normal_map = rt_texture_id_of(m.normalMap);
    r.is_anisotropic = m.isAnisotropic ? 1 : 0;
    r.alpha_x = m.alpha_x;
    r.alpha_y = m.alpha_y;
    r.has_img_texture = m.hasImgTexture ? 1 : 0;
    r.roughness = rt_texture_id_of(m.roughness);
    r.occlusion = rt_texture_id_of(m.occlusion);
    if (r.texture < 0 || r.normal_map < 0 || r.roughness < 0 || r.occlusion < 0)
        throw std::runtime_error("flatten: material points at a texture function the GPU renderer does not know");
    if (r.has_normal_map && r.normal_map == RT_TEX_NONE)
        throw std::runtime_error("flatten: hasNormalMap without a normalMap function");
    if (r.has_img_texture && (r.roughness == RT_TEX_NONE || r.occlusion == RT_TEX_NONE))
        throw std::runtime_error("flatten: hasImgTexture without roughness/occlusion functions");
    auto isBmp = [](int id) { return id >= RT_TEX_BMP_COLOR && id <= RT_TEX_BMP_OCCLUSION; };
    if (isBmp(r.texture) || isBmp(r.normal_map) || isBmp(r.roughness) || isBmp(r.occlusion)) uses_bmp = true;
    for (size_t k = 0; k < materials.size(); k++)
        if (std::memcmp(&materials[k], &r, sizeof r) == 0) return (int)k;
    materials.push_back(r);
    return (int)materials.size() - 1;
}

static void push3(std::vector<float> &v, const glm::vec3 &a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }
static void push2(std::vector<float> &v, const glm::vec2 &a) { v.push_back(a.x); v.push_back(a.y); }
static void pushMat(std::vector<float> &v, const glm::mat4 &m) {
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) v.push_back(m[c][r]);
}

void Sphere::flatten(FlatScene &out, int object_id) const {
    pushMat(out.sph_M, transformationMatrix);
    pushMat(out.sph_Minv, inverseTransformationMatrix);
    pushMat(out.sph_N, normalMatrix);
    out.sph_object.push_back(object_id);
}
void Cone::flatten(FlatScene &out, int object_id) const {
    pushMat(out.cone_M, transformationMatrix);
    pushMat(out.cone_Minv, inverseTransformationMatrix);
    pushMat(out.cone_N, normalMatrix);
    out.cone_object.push_back(object_id);
}
void Plane::flatten(FlatScene &out, int object_id) const {
    push3(out.pl_point, point);
    push3(out.pl_normal, normal);
    out.pl_object.push_back(object_id);
}
void MeshLoader::flatten(FlatScene &out, int object_id) const {
    for (const Triangle &t : triangles) {
        push3(out.v0, t.vertexA); push3(out.v1, t.vertexB); push3(out.v2, t.vertexC);
        push3(out.nA, t.normalA); push3(out.nB, t.normalB); push3(out.nC, t.normalC);
        push2(out.tA, t.textureA); push2(out.tB, t.textureB); push2(out.tC, t.textureC);
        out.tri_flags.push_back((t.vertexNormals ? 1u : 0u) | (t.vertexTextures ? 2u : 0u));
        out.tri_object.push_back(object_id);
    }
}

FlatScene flattenScene(const std::vector<Object *> &objs, const std::vector<Light *> &ls, glm::vec3 ambient) {
    FlatScene s;
    for (size_t o = 0; o < objs.size(); o++) {
        s.object_material.push_back(s.addMaterial(objs[o]->material));
        objs[o]->flatten(s, (int)o);
    }
    for (const Light *l : ls) {
        push3(s.light_pos, l->position);
        push3(s.light_color, l->color);
    }
    s.ambient[0] = ambient.x; s.ambient[1] = ambient.y; s.ambient[2] = ambient.z;
    return s;
}

void uploadScene(rt_context *ctx, const FlatScene &s, const std::string &texture_dir, bool commit) {
    const double t_set0 = ingestNowMs();
    rtCheck(rt_set_materials(ctx, (int32_t)s.materials.size(), s.materials.data()), "rt_set_materials");
    rtCheck(rt_set_object_materials(ctx, (int32_t)s.object_material.size(), s.object_material.data()), "rt_set_object_materials");
    rtCheck(rt_set_triangles(ctx, s.triangleCount(), s.v0.data(), s.v1.data(), s.v2.data(), s.nA.data(), s.nB.data(),
                             s.nC.data(), s.tA.data(), s.tB.data(), s.tC.data(), s.tri_flags.data(), s.tri_object.data()),
            "rt_set_triangles");
    rtCheck(rt_set_spheres(ctx, (int32_t)s.sph_object.size(), s.sph_M.data(), s.sph_Minv.data(), s.sph_N.data(), s.sph_object.data()), "rt_set_spheres");
    rtCheck(rt_set_cones(ctx, (int32_t)s.cone_object.size(), s.cone_M.data(), s.cone_Minv.data(), s.cone_N.data(), s.cone_object.data()), "rt_set_cones");
    rtCheck(rt_set_planes(ctx, (int32_t)s.pl_object.size(), s.pl_point.data(), s.pl_normal.data(), s.pl_object.data()), "rt_set_planes");
    rtCheck(rt_set_lights(ctx, (int32_t)(s.light_pos.size() / 3), s.light_pos.data(), s.light_color.data(), s.ambient), "rt_set_lights");
    lastIngest().ms_set = ingestNowMs() - t_set0;
    if (s.uses_bmp) {  // Textures.h:168-171
        const double t0 = ingestNowMs();
        const char *names[4] = {"basecolor.bmp", "normal.bmp", "roughness.bmp", "ambientOcclusion.bmp"};
        for (int slot = 0; slot < 4; slot++) {
            BmpImage img = readBmp(texture_dir + "/" + names[slot]);
            rtCheck(rt_set_bmp(ctx, slot, img.width, img.height, img.channels, img.pixels.data()), "rt_set_bmp");
        }
        lastIngest().ms_bmp = ingestNowMs() - t0;
    }
    if (commit) {
        rtCheck(rt_commit(ctx), "rt_commit");
        rtCheck(rt_get_commit_stats(ctx, &lastIngest().commit), "rt_get_commit_stats");
    }
}

void printIngest(std::FILE *f) {
    const IngestTimings &t = lastIngest();
    std::fprintf(f, "ingest: %d .obj files parsed in %.1f ms, flatten %.1f ms, BMP decode %.1f ms, rt_set_* %.1f ms, BVH build %.1f ms "
                    "(%llu triangles -> %llu nodes), reference order %.1f ms, records %.1f ms, upload %.1f ms\n",
                 t.meshes, t.ms_parse, t.ms_flatten, t.ms_bmp, t.ms_set, t.commit.ms_bvh_build, (unsigned long long)t.commit.n_triangles,
                 (unsigned long long)t.commit.bvh_nodes, t.commit.ms_reference_order, t.commit.ms_records, t.commit.ms_upload);
}

// the process-wide single-GPU context; `device` >= 0 names the GPU (else RT_DEVICE, else 0).  Asking for another GPU than
// the one the context lives on replaces it (the scene is uploaded again by the next render()/trace_ray()).
rt_context *defaultContext(int device) {
    static rt_context *ctx = nullptr;
    static int ctx_device = -1;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (device < 0 && !ctx) {
        const char *e = std::getenv("RT_DEVICE");
        device = e ? std::atoi(e) : 0;
    }
    if (ctx && device >= 0 && device != ctx_device) {
        if (g_tex_ctx == ctx) g_tex_ctx = nullptr;
        rt_destroy(ctx);
        ctx = nullptr;
        invalidateScene();
    }
    if (!ctx) {
        rtCheck(rt_create(device, &ctx), "rt_create");
        ctx_device = device;
    }
    return ctx;
}

// ---- render()/trace_ray(): the GPU stands where the tile loop stood ----------------------------------
// The uploaded scene is keyed on a fingerprint of everything the flattening reads: the object list (identity and
// order), every object's material and transformation, its geometry key (plane: point + normal; mesh: triangle
// storage + count + first/last triangle), the lights, ambient_light and the texture directory.  Editing any of
// these between two render()/trace_ray() calls re-uploads; in-place edits of a mesh's inner triangles (which the
// surface does not offer) need invalidateScene().
static uint64_t g_uploaded_print = 0;
static bool g_uploaded = false;
static std::string g_texture_dir = "./textures";
namespace {
struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t k = 0; k < n; k++) h = (h ^ b[k]) * 1099511628211ull;
    }
    template <class T> void pod(const T &v) { bytes(&v, sizeof v); }
};
}  // namespace
uint64_t Plane::geometryKey() const {
    Fnv f;
    f.pod(point.x); f.pod(point.y); f.pod(point.z);
    f.pod(normal.x); f.pod(normal.y); f.pod(normal.z);
    return f.h;
}
uint64_t MeshLoader::geometryKey() const {
    Fnv f;
    const size_t n = triangles.size();
    const void *store = (const void *)triangles.data();
    f.pod(n);
    f.pod(store);
    if (n) {
        for (const Triangle *t : {&triangles.front(), &triangles.back()}) {
            f.pod(t->vertexA.x); f.pod(t->vertexA.y); f.pod(t->vertexA.z);
            f.pod(t->vertexC.x); f.pod(t->vertexC.y); f.pod(t->vertexC.z);
        }
    }
    return f.h;
}
static uint64_t sceneFingerprint(const std::string &texture_dir) {
    Fnv f;
    FlatScene mats;  // materials go through the same conversion as the upload (function pointers -> texture ids)
    f.pod(objects.size());
    for (const Object *o : objects) {
        const void *id = (const void *)o;
        f.pod(id);
        const int m = mats.addMaterial(o->material);
        f.pod(m);
        const glm::mat4 &t = o->getTransformation();
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) f.pod(t[c][r]);
        f.pod(o->geometryKey());
    }
    for (const rt_material &m : mats.materials) f.pod(m);
    f.pod(lights.size());
    for (const Light *l : lights) {
        f.pod(l->position.x); f.pod(l->position.y); f.pod(l->position.z);
        f.pod(l->color.x); f.pod(l->color.y); f.pod(l->color.z);
    }
    f.pod(ambient_light.x); f.pod(ambient_light.y); f.pod(ambient_light.z);
    f.bytes(texture_dir.data(), texture_dir.size());
    return f.h;
}
void invalidateScene() { g_uploaded = false; }
void setTextureDir(const std::string &dir) { g_texture_dir = dir; }
// the renderer behind render()/trace_ray(): one context, or -- RenderOptions::devices with more than one entry -- an
// rt_multi over those GPUs (its rank 0 context also serves trace_ray() and the texture functions)
static rt_multi *g_multi = nullptr;
static std::vector<int> g_multi_devices;
static rt_context *rendererContext(const std::vector<int> &devices) {
    if (devices.size() <= 1) {
        if (g_multi) {  // back from several GPUs to one: the multi-GPU set is given up, the scene goes to the single context
            if (g_tex_ctx == rt_multi_context(g_multi, 0)) g_tex_ctx = nullptr;
            rt_multi_destroy(g_multi);
            g_multi = nullptr;
            g_multi_devices.clear();
            g_uploaded = false;
        }
        return defaultContext(devices.size() == 1 ? devices[0] : -1);
    }
    if (devices.size() > 1 && (!g_multi || devices != g_multi_devices)) {
        if (g_multi) rt_multi_destroy(g_multi);
        g_multi = nullptr;
        std::vector<int32_t> ids(devices.begin(), devices.end());
        rtCheck(rt_multi_create(ids.data(), (int32_t)ids.size(), 0, &g_multi), "rt_multi_create");
        g_multi_devices = devices;
        g_uploaded = false;
    }
    return rt_multi_context(g_multi, 0);
}
static void ensureUploaded(rt_context *ctx, const std::string &texture_dir) {
    const uint64_t print = sceneFingerprint(texture_dir);
    if (g_uploaded && print == g_uploaded_print) return;
    const double t_f0 = ingestNowMs();
    FlatScene s = flattenScene(objects, lights, ambient_light);
    lastIngest().ms_flatten = ingestNowMs() - t_f0;
    const bool multi = g_multi && ctx == rt_multi_context(g_multi, 0);
    uploadScene(ctx, s, texture_dir, !multi);
    if (multi) {
        rtCheck(rt_multi_commit(g_multi), "rt_multi_commit");  // BVH built once, uploaded to every GPU
        rtCheck(rt_get_commit_stats(ctx, &lastIngest().commit), "rt_get_commit_stats");
    }
    if (const char *v = std::getenv("RT_VERBOSE"))
        if (std::atoi(v) > 0) printIngest(stderr);
    rt_set_host_texture_context(ctx);
    g_uploaded_print = print;
    g_uploaded = true;
}

rt_stats render(Image &image, const Camera &camera, const RenderOptions &opt) {
    rt_context *ctx = rendererContext(opt.devices);
    g_texture_dir = opt.textureDir;  // trace_ray()/trace_rays() after a render() read the same maps
    ensureUploaded(ctx, opt.textureDir);
    rt_render_params p{};
    p.width = image.getWidth();
    p.height = image.getHeight();
    p.spp = opt.spp;
    p.max_depth = opt.maxDepth;
    p.seed = opt.seed;
    p.jitter_mode = opt.jitterMode;
    p.tile_rank = 0;
    p.tile_world = 1;
    p.flags = opt.singleFrame ? RT_RENDER_SINGLE_FRAME : 0;
    rt_camera cam = camera.toC();
    std::vector<uint8_t> rgb8((size_t)3 * p.width * p.height);
    rt_stats st{};
    if (g_multi && ctx == rt_multi_context(g_multi, 0)) {
        // N GPUs: interleaved tiles, one RCCL exchange at frame end (include/rt_hip.h: rt_multi_render)
        std::vector<rt_stats> per(rt_multi_size(g_multi));
        rtCheck(rt_multi_render(g_multi, &cam, &p, rgb8.data(), nullptr, nullptr, per.data()), "rt_multi_render");
        st = per[0];
        for (size_t k = 1; k < per.size(); k++) {  // every count adds up over the ranks, the frame takes as long as the slowest
            const rt_stats &o = per[k];
            st.primary_samples += o.primary_samples; st.rays_closest += o.rays_closest; st.rays_shadow += o.rays_shadow;
            st.shadow_culled += o.shadow_culled; st.shadow_zero += o.shadow_zero; st.shadow_unoccluded += o.shadow_unoccluded;
            st.hits_shaded += o.hits_shaded; st.queue_overflow += o.queue_overflow; st.split_items += o.split_items;
            st.nodes_visited += o.nodes_visited; st.tris_tested += o.tris_tested;
            st.nodes_visited_any += o.nodes_visited_any; st.tris_tested_any += o.tris_tested_any;
            st.chunks += o.chunks; st.chunk_retries += o.chunk_retries;
            st.launches_closest += o.launches_closest; st.launches_any += o.launches_any;
            st.max_walk_closest = std::max(st.max_walk_closest, o.max_walk_closest);
            st.max_walk_any = std::max(st.max_walk_any, o.max_walk_any);
            st.ms_total = std::max(st.ms_total, o.ms_total);
        }
    } else {
        rtCheck(rt_render(ctx, &cam, &p, rgb8.data(), nullptr, nullptr, &st), "rt_render");
    }
    image.setAll(rgb8.data());
    return st;
}

std::vector<glm::vec3> trace_rays(const std::vector<Ray> &rays, int bounces, std::vector<Hit> *hits) {
    rt_context *ctx = g_multi ? rt_multi_context(g_multi, 0) : defaultContext();  // (the renderer of the last render())
    ensureUploaded(ctx, g_texture_dir);  // the directory of the last render(), or setTextureDir(); default ./textures
    const size_t n = rays.size();
    std::vector<float> in(6 * n), col(3 * n);
    for (size_t k = 0; k < n; k++) {
        in[6 * k + 0] = rays[k].origin.x; in[6 * k + 1] = rays[k].origin.y; in[6 * k + 2] = rays[k].origin.z;
        in[6 * k + 3] = rays[k].direction.x; in[6 * k + 4] = rays[k].direction.y; in[6 * k + 5] = rays[k].direction.z;
    }
    std::vector<rt_hit> h(hits ? n : 0);
    rtCheck(rt_trace_batch(ctx, (int64_t)n, in.data(), bounces, col.data(), hits ? h.data() : nullptr, nullptr, nullptr), "rt_trace_batch");
    std::vector<glm::vec3> out(n);
    for (size_t k = 0; k < n; k++) out[k] = glm::vec3(col[3 * k], col[3 * k + 1], col[3 * k + 2]);
    if (hits) {
        hits->resize(n);
        for (size_t k = 0; k < n; k++) {
            Hit &d = (*hits)[k];
            const rt_hit &s = h[k];
            d.hit = s.hit != 0;
            d.distance = s.distance;
            d.object = (s.object >= 0 && (size_t)s.object < objects.size()) ? objects[s.object] : nullptr;
            d.triangle = s.triangle;
            d.intersection = glm::vec3(s.intersection[0], s.intersection[1], s.intersection[2]);
            d.normal = glm::vec3(s.normal[0], s.normal[1], s.normal[2]);
            d.normalShading = glm::vec3(s.normal_shading[0], s.normal_shading[1], s.normal_shading[2]);
            d.uv = glm::vec2(s.uv[0], s.uv[1]);
            d.tangent = glm::vec3(s.tangent[0], s.tangent[1], s.tangent[2]);
            d.bitangent = glm::vec3(s.bitangent[0], s.bitangent[1], s.bitangent[2]);
        }
    }
    return out;
}

glm::vec3 trace_ray(Ray ray, int bounces) { return trace_rays({ray}, bounces)[0]; }

void clearScene() {
    for (Object *o : objects) delete o;
    for (Light *l : lights) delete l;
    objects.clear();
    lights.clear();
    ambient_light = glm::vec3(0.7f);
    g_uploaded = false;
    lastIngest() = IngestTimings{};
}
