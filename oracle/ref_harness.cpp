// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never measured as the product).
//
// Build-owned driver around the *unmodified* reference sources. The reference translation unit
// (/root/reference/main.cpp and the headers it pulls in) is #included where it lies; nothing from it
// is copied into this repository. Its main() is renamed so that this file can drive
// sampleScene()/competitionScene(), trace_ray(), closest(), is_shadowed(), PhongModel(), the texture
// functions and MeshLoader with run-time parameters and dump fixtures (tests/golden/*, see
// oracle/make_golden.py).  Built only in the authoring container by oracle/Makefile into oracle/_ref/.
//
// Why a harness and not the shipped main(): SURVEY.md F3/F4 -- meshes/pietre.obj is a missing blob, the
// shipped main() segfaults at 0.23 %; the harness erases that dead object after competitionScene().
//
// Output format: raw little-endian arrays, described by the `dump_*` helpers below; make_golden.py reads them.
#include <iostream>
#include <fstream>
#include <sstream>
#include <vector>
#include <string>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <algorithm>
#include <stdexcept>
#include <omp.h>

// Open the reference's private members to the harness (triangle lists, primitive parameters).
#define private public
#define protected public
#define main reference_main
#include "/root/reference/main.cpp"
#undef main
#undef private
#undef protected

// ---------------------------------------------------------------------------------------------
// shared counter-based jitter (identical formula in oracle/rt_oracle.cpp and the HIP kernels)
static inline uint32_t rt_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static inline void rt_jitter(uint32_t pixel, uint32_t sample, uint32_t seed, float &jx, float &jy) {
    uint32_t a = rt_hash32(pixel ^ (seed * 0x85ebca6bU + 0x9e3779b9U));
    uint32_t b = rt_hash32(a + 2u * sample + 1u);
    uint32_t c = rt_hash32(a + 2u * sample + 2u);
    jx = ((float)(b >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1.5f;
    jy = ((float)(c >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1.5f;
}
static const float kJitter4[8] = {-1.0 / 4.0, 3.0 / 4.0, 3.0 / 4.0, 1.0 / 3.0,
                                  -3.0 / 4.0, -1.0 / 4.0, 1.0 / 4.0, -3.0 / 4.0};

struct Camera {
    glm::vec3 origin;
    glm::mat4 rot;
    float dz;
};

static Camera sample_camera() { return {glm::vec3(0.0), glm::mat4(1.0f), 1.0f}; }
static Camera competition_camera() {
    glm::mat4 r = glm::rotate(glm::mat4(1.0f), -0.75f, glm::vec3(1.0f, 0.0f, 0.0f));
    r = glm::rotate(r, 0.4f, glm::vec3(0.0f, 1.0f, 0.0f));
    return {glm::vec3(-0.45, -0.21, 1.52), r, -1.0f};
}

// a scene for the Cone primitive (Objects.h:188-278), which neither shipped scene instantiates.
// Defined here with the reference's own classes; raytracer_amd/host/Scenes.cpp (coneScene) restates it through the
// kept surface.  Cones: apex at the object-space origin, opening towards +y, cap at y = 1.
static void cones_scene() {
    Material matte, mirror, glass, perla, brushed, ground, wall;
    matte.ambient = glm::vec3(0.02f, 0.02f, 0.05f);
    matte.diffuse = glm::vec3(0.3f, 0.4f, 0.9f);
    matte.specular = glm::vec3(0.3f);
    matte.shininess = 10.0f;
    mirror.reflection = 0.8f;
    mirror.diffuse = glm::vec3(0.1f);
    mirror.specular = glm::vec3(0.5f);
    mirror.shininess = 50.0f;
    glass.diffuse = glm::vec3(0.0f);
    glass.refraction = 1.0f;
    glass.reflection = 1.0f;
    glass.sigma = 1.5f;
    perla.ambient = glm::vec3(0.05f);
    perla.texture = &opal;
    perla.refraction = 0.5f;
    perla.reflection = 0.1f;
    perla.sigma = 2.0f;
    brushed.ambient = glm::vec3(0.01f, 0.03f, 0.03f);
    brushed.diffuse = glm::vec3(0.8f, 0.4f, 0.1f);
    brushed.specular = glm::vec3(0.6f);
    brushed.isAnisotropic = true;
    ground.ambient = glm::vec3(0.03f);
    ground.texture = &perlinTerrain;
    wall.diffuse = glm::vec3(0.6f, 0.6f, 0.5f);
    wall.specular = glm::vec3(0.2f);
    wall.shininess = 20.0f;
    wall.hasNormalMap = true;
    wall.normalMap = &perlinNormal;

    objects.push_back(new MeshLoader("./meshes/bunny.obj", glm::vec3(-1.0f, -3.0f, 6.5f), true, matte));
    objects.push_back(new Plane(glm::vec3(0.0f, -3.0f, 0.0f), glm::vec3(0.0f, 1.0f, 0.0f), true, ground));
    objects.push_back(new Plane(glm::vec3(0.0f, 0.0f, 18.0f), glm::vec3(0.0f, 0.0f, -1.0f), true, wall));
    objects.push_back(new Plane(glm::vec3(-9.0f, 0.0f, 0.0f), glm::vec3(1.0f, 0.0f, 0.0f), true, matte));
    struct { const Material *m; glm::vec3 at; float angle; glm::vec3 axis; glm::vec3 scale; } cones[6] = {
        {&matte, glm::vec3(-3.5f, -3.0f, 9.0f), 0.0f, glm::vec3(1, 0, 0), glm::vec3(1.5f, 3.0f, 1.5f)},              // apex down
        {&mirror, glm::vec3(2.5f, 0.5f, 11.0f), 3.14159265f, glm::vec3(1, 0, 0), glm::vec3(1.4f, 3.5f, 1.4f)},      // upright
        {&glass, glm::vec3(0.3f, -1.2f, 6.0f), 0.6f, glm::vec3(0, 0, 1), glm::vec3(0.8f, 1.6f, 0.8f)},              // tilted
        {&perla, glm::vec3(4.5f, -3.0f, 7.5f), 0.0f, glm::vec3(1, 0, 0), glm::vec3(1.0f, 2.0f, 1.0f)},
        {&brushed, glm::vec3(-4.5f, 1.5f, 13.0f), 2.5f, glm::vec3(1, 0, 1), glm::vec3(1.5f, 2.0f, 1.5f)},
        {&mirror, glm::vec3(0.0f, 3.0f, 14.0f), 1.5707963f, glm::vec3(0, 0, 1), glm::vec3(2.0f, 1.0f, 0.5f)},       // squashed
    };
    for (auto &c : cones) {
        Cone *cone = new Cone(*c.m);
        cone->setTransformation(glm::translate(c.at) * glm::rotate(glm::mat4(1.0f), c.angle, c.axis) * glm::scale(c.scale));
        objects.push_back(cone);
    }
    Sphere *s = new Sphere(mirror);
    s->setTransformation(glm::translate(glm::vec3(-6.0f, -1.5f, 12.0f)) * glm::scale(glm::vec3(1.5f)));
    objects.push_back(s);
    lights.push_back(new Light(glm::vec3(0.0f, 10.0f, 4.0f), glm::vec3(60.0f)));
    lights.push_back(new Light(glm::vec3(-6.0f, 3.0f, 2.0f), glm::vec3(25.0f)));
    lights.push_back(new Light(glm::vec3(5.0f, 1.0f, 0.0f), glm::vec3(15.0f)));
}

static Camera load_scene(const std::string &name) {
    if (name == "cones") {
        cones_scene();
        return sample_camera();
    }
    if (name == "sample") {
        sampleScene();
        return sample_camera();
    }
    if (name == "bunny") {  // sampleScene minus the 10 analytic spheres
        sampleScene();
        objects.erase(objects.begin() + 7, objects.end());
        return sample_camera();
    }
    if (name == "competition") {
        competitionScene();
        objects.erase(objects.begin() + 1);  // dead pietre.obj entry
        return competition_camera();
    }
    throw std::runtime_error("unknown scene " + name);
}

template <class T>
static void dump(std::ofstream &f, const std::vector<T> &v) {
    f.write(reinterpret_cast<const char *>(v.data()), v.size() * sizeof(T));
}

static Ray primary_ray(const Camera &cam, int width, int height, float fov, int i, int j, float jx, float jy) {
    const float s = 2 * tan(0.5 * fov / 180 * M_PI) / width;
    const float X = -s * width / 2;
    const float Y = s * height / 2;
    float dx = X + (i + jx) * s + s / 2;
    float dy = Y - (j + jy) * s - s / 2;
    glm::vec4 direction4(dx, dy, cam.dz, 0.0f);
    direction4 = cam.rot * direction4;
    glm::vec3 direction = glm::normalize(glm::vec3(direction4));
    return Ray(cam.origin, direction);
}

// (object index, triangle index within mesh or -1) of a closest hit; first triangle with the same distance
static void prim_of_hit(Ray ray, const Hit &h, int &obj, int &tri) {
    obj = -1;
    tri = -1;
    if (!h.hit) return;
    for (size_t k = 0; k < objects.size(); k++)
        if (objects[k] == h.object) obj = (int)k;
    MeshLoader *m = dynamic_cast<MeshLoader *>(h.object);
    if (!m) return;
    for (size_t t = 0; t < m->triangles.size(); t++) {
        Hit th = m->triangles[t].intersect(ray);
        if (th.hit && th.distance == h.distance) {
            tri = (int)t;
            return;
        }
    }
}

// ---------------------------------------------------------------------------------------------
static int cmd_render(int argc, char **argv) {
    // render <scene> <w> <h> <spp> <depth> <seed> <out_prefix> [ids]
    std::string scene = argv[2];
    int width = atoi(argv[3]), height = atoi(argv[4]), spp = atoi(argv[5]), depth = atoi(argv[6]);
    uint32_t seed = (uint32_t)atoll(argv[7]);
    std::string out = argv[8];
    bool want_ids = argc > 9 && std::string(argv[9]) == "ids";
    auto t0 = std::chrono::high_resolution_clock::now();
    Camera cam = load_scene(scene);
    auto t1 = std::chrono::high_resolution_clock::now();
    std::vector<int32_t> rgb8(3 * (size_t)width * height);
    std::vector<float> rgbf(3 * (size_t)width * height);
    std::vector<int32_t> ids(2 * (size_t)width * height, -1);
    const int tile_size = 16;
    const int tiles_x = (width + tile_size - 1) / tile_size;
    const int tiles_y = (height + tile_size - 1) / tile_size;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < tiles_x * tiles_y; tile++) {
        int tj = tile / tiles_x, ti = tile - tj * tiles_x;
        for (int i = ti * tile_size; i < std::min(ti * tile_size + tile_size, width); i++)
            for (int j = tj * tile_size; j < std::min(tj * tile_size + tile_size, height); j++) {
                glm::vec3 pixelColor(0.0f);
                for (int sample = 0; sample < spp; ++sample) {
                    float jx, jy;
                    if (spp == 4) {
                        jx = kJitter4[2 * sample];
                        jy = kJitter4[2 * sample + 1];
                    } else {
                        rt_jitter((uint32_t)(j * width + i), (uint32_t)sample, seed, jx, jy);
                    }
                    Ray ray = primary_ray(cam, width, height, 90, i, j, jx, jy);
                    pixelColor += trace_ray(ray, depth);
                    if (sample == 0 && want_ids) {
                        Hit h = closest(ray);
                        int o, t;
                        prim_of_hit(ray, h, o, t);
                        ids[2 * ((size_t)j * width + i)] = o;
                        ids[2 * ((size_t)j * width + i) + 1] = t;
                    }
                }
                pixelColor /= (float)spp;
                size_t p = 3 * ((size_t)j * width + i);
                rgbf[p] = pixelColor.x;
                rgbf[p + 1] = pixelColor.y;
                rgbf[p + 2] = pixelColor.z;
                glm::vec3 c = toneMapping(pixelColor);
                rgb8[p] = (float)(255 * c.r);  // Image::setPixel(vec3) truncation, Image.h:91-95
                rgb8[p + 1] = (float)(255 * c.g);
                rgb8[p + 2] = (float)(255 * c.b);
            }
    }
    auto t2 = std::chrono::high_resolution_clock::now();
    double load_s = std::chrono::duration<double>(t1 - t0).count();
    double render_s = std::chrono::duration<double>(t2 - t1).count();
    if (out != "-") {
        std::ofstream f(out + ".rgb8.i32", std::ios::binary);
        dump(f, rgb8);
        std::ofstream g(out + ".rgbf.f32", std::ios::binary);
        dump(g, rgbf);
        if (want_ids) {
            std::ofstream h(out + ".ids.i32", std::ios::binary);
            dump(h, ids);
        }
    }
    printf("{\"scene\": \"%s\", \"width\": %d, \"height\": %d, \"spp\": %d, \"depth\": %d, \"threads\": %d, "
           "\"load_s\": %.6f, \"render_s\": %.6f, \"primary_samples\": %lld}\n",
           scene.c_str(), width, height, spp, depth, omp_get_max_threads(), load_s, render_s,
           (long long)width * height * spp);
    return 0;
}

// ---------------------------------------------------------------------------------------------
static int cmd_perlin(int argc, char **argv) {
    // perlin <in_points.f32 (n*3)> <out.f32 (n)>
    std::ifstream in(argv[2], std::ios::binary);
    std::vector<char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t n = buf.size() / 12;
    const float *p = reinterpret_cast<const float *>(buf.data());
    std::vector<float> o(n);
    for (size_t k = 0; k < n; k++) o[k] = glm::perlin(glm::vec3(p[3 * k], p[3 * k + 1], p[3 * k + 2]));
    std::ofstream f(argv[3], std::ios::binary);
    dump(f, o);
    return 0;
}

static int cmd_textures(int argc, char **argv) {
    // textures <in_uv.f32 (n*2)> <out.f32 (11*n*3)> [tmax]  order: perlinNormal perlinWater opal perlinTerrain
    //          perlinIceTerrain snowTerrain qwilfishTexture colorAt normalAt roughnessAt ambientOcclusionAt
    std::ifstream in(argv[2], std::ios::binary);
    std::vector<char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t n = buf.size() / 8;
    const float *p = reinterpret_cast<const float *>(buf.data());
    typedef glm::vec3 (*tex3)(glm::vec2);
    tex3 fs[9] = {perlinNormal, perlinWater, opal, perlinTerrain, perlinIceTerrain, snowTerrain,
                  qwilfishTexture, colorAt, normalAt};
    const int tmax = argc > 4 ? atoi(argv[4]) : 11;  // evaluate only the first tmax functions (BMP fetches are UB outside [0,1))
    std::vector<float> o;
    for (int t = 0; t < 11; t++)
        for (size_t k = 0; k < n; k++) {
            glm::vec2 uv(p[2 * k], p[2 * k + 1]);
            glm::vec3 c(0.0f);
            if (t >= tmax) c = glm::vec3(0.0f);
            else if (t < 9) c = fs[t](uv);
            else if (t == 9) c = glm::vec3(roughnessAt(uv));
            else c = glm::vec3(ambientOcclusionAt(uv));
            o.push_back(c.x);
            o.push_back(c.y);
            o.push_back(c.z);
        }
    std::ofstream f(argv[3], std::ios::binary);
    dump(f, o);
    return 0;
}

// ---------------------------------------------------------------------------------------------
static int cmd_mesh(int argc, char **argv) {
    // mesh <objfile> <tx> <ty> <tz> <out.f32>: per triangle 29 floats:
    //   A(3) B(3) C(3) n(3) nA(3) nB(3) nC(3) tA(2) tB(2) tC(2) hasN hasT  + trailing 6 floats mesh box(min,max)
    Material m;
    MeshLoader ml(argv[2], glm::vec3(atof(argv[3]), atof(argv[4]), atof(argv[5])), true, m);
    std::vector<float> o;
    for (Triangle &t : ml.triangles) {
        auto p3 = [&](glm::vec3 v) { o.push_back(v.x); o.push_back(v.y); o.push_back(v.z); };
        auto p2 = [&](glm::vec2 v) { o.push_back(v.x); o.push_back(v.y); };
        p3(t.vertexA); p3(t.vertexB); p3(t.vertexC); p3(t.normal);
        p3(t.normalA); p3(t.normalB); p3(t.normalC);
        p2(t.textureA); p2(t.textureB); p2(t.textureC);
        o.push_back(t.vertexNormals ? 1.f : 0.f);
        o.push_back(t.vertexTextures ? 1.f : 0.f);
    }
    o.push_back(ml.boundingBox.minBounds.x); o.push_back(ml.boundingBox.minBounds.y); o.push_back(ml.boundingBox.minBounds.z);
    o.push_back(ml.boundingBox.maxBounds.x); o.push_back(ml.boundingBox.maxBounds.y); o.push_back(ml.boundingBox.maxBounds.z);
    std::ofstream f(argv[6], std::ios::binary);
    dump(f, o);
    printf("{\"triangles\": %zu, \"vertices\": %zu, \"normals\": %zu, \"uvs\": %zu}\n", ml.triangles.size(),
           ml.vertices.size(), ml.normals.size(), ml.textureCoords.size());
    return 0;
}

// ---------------------------------------------------------------------------------------------
static int cmd_rays(int argc, char **argv) {
    // rays <scene> <in_rays.f32 (n*6: origin, direction[already normalised])> <out.f32>
    // per ray 32 floats: hit obj tri distance  P(3) N(3) Ns(3) uv(2) T(3) B(3)  shadow[4]  color_d0(3) color_d3(3) pad(1)
    std::string scene = argv[2];
    load_scene(scene);
    std::ifstream in(argv[3], std::ios::binary);
    std::vector<char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t n = buf.size() / 24;
    const float *p = reinterpret_cast<const float *>(buf.data());
    std::vector<float> o(n * 32, 0.f);
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t k = 0; k < n; k++) {
        Ray ray(glm::vec3(p[6 * k], p[6 * k + 1], p[6 * k + 2]), glm::vec3(p[6 * k + 3], p[6 * k + 4], p[6 * k + 5]));
        Hit h = closest(ray);
        float *r = &o[32 * k];
        int obj, tri;
        prim_of_hit(ray, h, obj, tri);
        r[0] = h.hit ? 1.f : 0.f;
        r[1] = (float)obj;
        r[2] = (float)tri;
        r[3] = h.distance;
        if (h.hit) {
            r[4] = h.intersection.x; r[5] = h.intersection.y; r[6] = h.intersection.z;
            r[7] = h.normal.x; r[8] = h.normal.y; r[9] = h.normal.z;
            r[10] = h.normalShading.x; r[11] = h.normalShading.y; r[12] = h.normalShading.z;
            r[13] = h.uv.x; r[14] = h.uv.y;
            r[15] = h.tangent.x; r[16] = h.tangent.y; r[17] = h.tangent.z;
            r[18] = h.bitangent.x; r[19] = h.bitangent.y; r[20] = h.bitangent.z;
            // shadow test per light exactly as PhongModel calls it (after the normal flip, main.cpp:88-100)
            glm::vec3 view = glm::normalize(-ray.direction);
            glm::vec3 nrm = h.normal;
            if (glm::dot(nrm, view) < 0) nrm = -nrm;
            for (size_t l = 0; l < lights.size() && l < 4; l++) {
                glm::vec3 ld = glm::normalize(lights[l]->position - h.intersection);
                float dist = glm::distance(h.intersection, lights[l]->position);
                r[21 + l] = is_shadowed(h.intersection, nrm, ld, dist) ? 1.f : 0.f;
            }
        }
        glm::vec3 c0 = trace_ray(ray, 0), c3 = trace_ray(ray, 3);
        r[25] = c0.x; r[26] = c0.y; r[27] = c0.z;
        r[28] = c3.x; r[29] = c3.y; r[30] = c3.z;
    }
    std::ofstream f(argv[4], std::ios::binary);
    dump(f, o);
    return 0;
}

// ---------------------------------------------------------------------------------------------
static int cmd_scene(int argc, char **argv) {
    // scene <scene> <out.f32>: dump analytic primitives + lights so the host surface can be checked
    // spheres (tag 1) and cones (tag 4): 48 floats (M, Minv, Nmat column-major); planes: 6 floats (point, normal)
    std::string scene = argv[2];
    load_scene(scene);
    std::vector<float> o;
    for (Object *ob : objects) {
        if (Sphere *s = dynamic_cast<Sphere *>(ob)) {
            o.push_back(1.f);
            const glm::mat4 *ms[3] = {&s->transformationMatrix, &s->inverseTransformationMatrix, &s->normalMatrix};
            for (auto m : ms)
                for (int c = 0; c < 4; c++)
                    for (int r = 0; r < 4; r++) o.push_back((*m)[c][r]);
        } else if (Cone *cn = dynamic_cast<Cone *>(ob)) {
            o.push_back(4.f);
            const glm::mat4 *ms[3] = {&cn->transformationMatrix, &cn->inverseTransformationMatrix, &cn->normalMatrix};
            for (auto m : ms)
                for (int c = 0; c < 4; c++)
                    for (int r = 0; r < 4; r++) o.push_back((*m)[c][r]);
        } else if (Plane *p = dynamic_cast<Plane *>(ob)) {
            o.push_back(2.f);
            o.push_back(p->point.x); o.push_back(p->point.y); o.push_back(p->point.z);
            o.push_back(p->normal.x); o.push_back(p->normal.y); o.push_back(p->normal.z);
        } else if (MeshLoader *m = dynamic_cast<MeshLoader *>(ob)) {
            o.push_back(3.f);
            o.push_back((float)m->triangles.size());
        }
    }
    std::ofstream f(argv[3], std::ios::binary);
    dump(f, o);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: ref_harness render|perlin|textures|mesh|rays|scene ...\n");
        return 2;
    }
    std::string c = argv[1];
    try {
        if (c == "render" && argc >= 9) return cmd_render(argc, argv);
        if (c == "perlin" && argc >= 4) return cmd_perlin(argc, argv);
        if (c == "textures" && argc >= 4) return cmd_textures(argc, argv);
        if (c == "mesh" && argc >= 7) return cmd_mesh(argc, argv);
        if (c == "rays" && argc >= 5) return cmd_rays(argc, argv);
        if (c == "scene" && argc >= 4) return cmd_scene(argc, argv);
    } catch (std::exception &e) {
        fprintf(stderr, "ref_harness: %s\n", e.what());
        return 1;
    }
    fprintf(stderr, "ref_harness: bad arguments\n");
    return 2;
}
