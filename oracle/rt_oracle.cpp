// rt_oracle.cpp -- CPU ORACLE: a restatement of the reference's per-pixel trace loop.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Nothing in the product links against or imports this file.
//
// Every function names the reference lines it follows (paths relative to /root/reference).  The goal is
// bit-for-bit agreement with the reference built `g++ -O2 -fopenmp` (the IEEE golden), so:
//   * float expressions keep the reference's (and glm 0.9.9.7's) operand order and mixed float/double
//     promotions (checked with static_asserts against the real headers while writing this file);
//   * this file is compiled -O2 -ffp-contract=off -fno-fast-math;
//   * the per-mesh BVH is the reference's own (mean split, leaf <= 100, its slab test), because its candidate
//     ORDER decides exact distance ties.
// What is deliberately different: no std::vector<Triangle> copies during traversal -- the
// candidate leaves are visited in the same order without materialising them, which changes nothing observable.
// Pin status: pinned against oracle/_ref (the compiled reference) through tests/golden/* -- tests/test_oracle_pin.py.
#include "rt_oracle.h"

#include <omp.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------------------
// vector helpers in glm operand order
struct V2 { float x, y; };
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3s(float s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline V2 operator*(float s, V2 a) { return V2{s * a.x, s * a.y}; }
inline V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
// glm/detail/func_geometric.inl:48-55
inline float dot(V3 a, V3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
// glm/detail/func_geometric.inl:70-78
inline V3 cross(V3 x, V3 y) { return V3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline float length(V3 v) { return std::sqrt(dot(v, v)); }                     // :10-13
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }        // :82-90 + func_exponential.inl inversesqrt
inline float gmin(float x, float y) { return (y < x) ? y : x; }                // glm::min
inline float gmax(float x, float y) { return (x < y) ? y : x; }                // glm::max
inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
inline float smax(float a, float b) { return (a < b) ? b : a; }                // std::max(a, b)

struct M4 { float c[4][4]; };  // column-major, c[col][row]
// glm/detail/type_mat4x4.inl:561-572 -> xyz of (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
inline V3 mulPoint(const M4 &m, V3 v, float w) {
    float r[3];
    for (int k = 0; k < 3; k++) {
        float a0 = m.c[0][k] * v.x + m.c[1][k] * v.y;
        float a1 = m.c[2][k] * v.z + m.c[3][k] * w;
        r[k] = a0 + a1;
    }
    return V3{r[0], r[1], r[2]};
}

// ---------------------------------------------------------------------------------------------------------
// glm::perlin(vec3): glm/gtc/noise.inl:65-132 with glm/detail/_noise.hpp:8-78 and func_common.inl (floor, fract,
// step, mix, abs)
inline float mod289(float x) { return x - std::floor(x * (1.0f / 289.0f)) * 289.0f; }
inline float permute(float x) { return mod289(((x * 34.0f) + 1.0f) * x); }
inline float fractf_(float x) { return x - std::floor(x); }
inline float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
inline float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
inline float fade(float t) { return (t * t * t) * (t * (t * 6.0f - 15.0f) + 10.0f); }
inline float taylorInvSqrt(float r) { return 1.79284291400159f - 0.85373472095314f * r; }

float perlin3(float px, float py, float pz) {
    const float P[3] = {px, py, pz};
    float Pi0[3], Pi1[3], Pf0[3], Pf1[3];
    for (int k = 0; k < 3; k++) {
        float fl = std::floor(P[k]);
        Pi0[k] = mod289(fl);
        Pi1[k] = mod289(fl + 1.0f);
        Pf0[k] = P[k] - std::floor(P[k]);
        Pf1[k] = Pf0[k] - 1.0f;
    }
    const float ix[4] = {Pi0[0], Pi1[0], Pi0[0], Pi1[0]};
    const float iy[4] = {Pi0[1], Pi0[1], Pi1[1], Pi1[1]};
    float gx0[4], gy0[4], gz0[4], gx1[4], gy1[4], gz1[4];
    const float seventh = (float)(1.0 / 7.0);
    for (int k = 0; k < 4; k++) {
        float ixy = permute(permute(ix[k]) + iy[k]);
        float ixy0 = permute(ixy + Pi0[2]);
        float ixy1 = permute(ixy + Pi1[2]);
        {
            float gx = ixy0 * seventh;
            float gy = fractf_(std::floor(gx) * seventh) - 0.5f;
            gx = fractf_(gx);
            float gz = 0.5f - std::fabs(gx) - std::fabs(gy);
            float sz = stepf(gz, 0.0f);
            gx -= sz * (stepf(0.0f, gx) - 0.5f);
            gy -= sz * (stepf(0.0f, gy) - 0.5f);
            gx0[k] = gx; gy0[k] = gy; gz0[k] = gz;
        }
        {
            float gx = ixy1 * seventh;
            float gy = fractf_(std::floor(gx) * seventh) - 0.5f;
            gx = fractf_(gx);
            float gz = 0.5f - std::fabs(gx) - std::fabs(gy);
            float sz = stepf(gz, 0.0f);
            gx -= sz * (stepf(0.0f, gx) - 0.5f);
            gy -= sz * (stepf(0.0f, gy) - 0.5f);
            gx1[k] = gx; gy1[k] = gy; gz1[k] = gz;
        }
    }
    // lane k of the 4-vectors: 0 -> (x0,y0), 1 -> (x1,y0), 2 -> (x0,y1), 3 -> (x1,y1)
    V3 g000 = v3(gx0[0], gy0[0], gz0[0]), g100 = v3(gx0[1], gy0[1], gz0[1]);
    V3 g010 = v3(gx0[2], gy0[2], gz0[2]), g110 = v3(gx0[3], gy0[3], gz0[3]);
    V3 g001 = v3(gx1[0], gy1[0], gz1[0]), g101 = v3(gx1[1], gy1[1], gz1[1]);
    V3 g011 = v3(gx1[2], gy1[2], gz1[2]), g111 = v3(gx1[3], gy1[3], gz1[3]);
    g000 = g000 * taylorInvSqrt(dot(g000, g000));
    g010 = g010 * taylorInvSqrt(dot(g010, g010));
    g100 = g100 * taylorInvSqrt(dot(g100, g100));
    g110 = g110 * taylorInvSqrt(dot(g110, g110));
    g001 = g001 * taylorInvSqrt(dot(g001, g001));
    g011 = g011 * taylorInvSqrt(dot(g011, g011));
    g101 = g101 * taylorInvSqrt(dot(g101, g101));
    g111 = g111 * taylorInvSqrt(dot(g111, g111));
    float n000 = dot(g000, v3(Pf0[0], Pf0[1], Pf0[2]));
    float n100 = dot(g100, v3(Pf1[0], Pf0[1], Pf0[2]));
    float n010 = dot(g010, v3(Pf0[0], Pf1[1], Pf0[2]));
    float n110 = dot(g110, v3(Pf1[0], Pf1[1], Pf0[2]));
    float n001 = dot(g001, v3(Pf0[0], Pf0[1], Pf1[2]));
    float n101 = dot(g101, v3(Pf1[0], Pf0[1], Pf1[2]));
    float n011 = dot(g011, v3(Pf0[0], Pf1[1], Pf1[2]));
    float n111 = dot(g111, v3(Pf1[0], Pf1[1], Pf1[2]));
    float fx = fade(Pf0[0]), fy = fade(Pf0[1]), fz = fade(Pf0[2]);
    float nz0 = mixf(n000, n001, fz), nz1 = mixf(n100, n101, fz), nz2 = mixf(n010, n011, fz), nz3 = mixf(n110, n111, fz);
    float nyz0 = mixf(nz0, nz2, fy), nyz1 = mixf(nz1, nz3, fy);
    float nxyz = mixf(nyz0, nyz1, fx);
    return 2.2f * nxyz;
}

// ---------------------------------------------------------------------------------------------------------
struct Scene;
V3 evalTexture(const Scene &s, int id, V2 uv);

struct Tri {
    V3 A, B, C, n;
    V3 nA, nB, nC;
    V2 tA, tB, tC;
    bool vertexNormals, vertexTextures;
};

// BoundingBox, reference MeshLoader.h:127-209
struct Box {
    V3 mn = v3s(INFINITY), mx = v3s(-INFINITY);
    void grow(V3 p) {
        mn.x = std::min(mn.x, p.x); mn.y = std::min(mn.y, p.y); mn.z = std::min(mn.z, p.z);
        mx.x = std::max(mx.x, p.x); mx.y = std::max(mx.y, p.y); mx.z = std::max(mx.z, p.z);
    }
    // MeshLoader.h:151-191 (only .hit is ever used by callers)
    bool hit(V3 o, V3 d) const {
        float tmin = (mn.x - o.x) / d.x, tmax = (mx.x - o.x) / d.x;
        if (tmin > tmax) std::swap(tmin, tmax);
        float ymin = (mn.y - o.y) / d.y, ymax = (mx.y - o.y) / d.y;
        if (ymin > ymax) std::swap(ymin, ymax);
        if ((tmin > ymax) || (ymin > tmax)) return false;
        if (ymin > tmin) tmin = ymin;
        if (ymax < tmax) tmax = ymax;
        float zmin = (mn.z - o.z) / d.z, zmax = (mx.z - o.z) / d.z;
        if (zmin > zmax) std::swap(zmin, zmax);
        if ((tmin > zmax) || (zmin > tmax)) return false;
        float t_enter = std::max(std::max(tmin, ymin), zmin);
        float t_exit = std::min(std::min(tmax, ymax), zmax);
        if (t_enter > t_exit || t_exit < 0) return false;
        return true;
    }
};

// bvh_node, reference MeshLoader.h:211-285
struct RefNode {
    Box box;
    int left = -1, right = -1;
    std::vector<int> tris;  // leaf: global triangle indices in file order
};

struct Mesh {
    int object = -1;
    Box box;                 // MeshLoader::boundingBox (here from the triangles)
    std::vector<RefNode> nodes;  // nodes[0] = root
    int64_t first = 0, count = 0;
};

struct Sphere { M4 M, Minv, N; int object; };
struct Plane { V3 point, normal; int object; };
struct Light { V3 pos, color; };

struct Hit {
    bool hit = false;
    V3 normal{0, 0, 0}, P{0, 0, 0};
    float distance = 0;
    int object = -1;
    int64_t tri = -1;  // global triangle index
    V2 uv{0, 0};
    V3 ns{0, 0, 0}, T{0, 0, 0}, B{0, 0, 0};
};

struct Counters {
    uint64_t closest = 0, shadow = 0, culled = 0, shaded = 0;
};

struct Scene {
    std::vector<rt_material> materials;
    std::vector<int32_t> object_material;
    std::vector<Tri> tris;
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<Sphere> cones;  // same record: three matrices + object
    std::vector<Plane> planes;
    std::vector<Light> lights;
    V3 ambient{0, 0, 0};
    int bw[4] = {0, 0, 0, 0}, bh[4] = {0, 0, 0, 0}, bc[4] = {0, 0, 0, 0};
    std::vector<uint8_t> bmp[4];
    // scene order: kind 0 mesh, 1 sphere, 2 plane, 3 cone; idx into the respective vector
    struct Ref { int kind, idx; };
    std::vector<Ref> order;
};

// splitMesh + ctor, MeshLoader.h:218-263
int buildRef(const Scene &s, Mesh &m, const std::vector<int> &mesh, int a, int depth) {
    int id = (int)m.nodes.size();
    m.nodes.emplace_back();
    for (int t : mesh) {
        const Tri &T = s.tris[t];
        m.nodes[id].box.grow(T.A);
        m.nodes[id].box.grow(T.B);
        m.nodes[id].box.grow(T.C);
    }
    if (mesh.size() <= 100 || depth > 48) {
        m.nodes[id].tris = mesh;
        return id;
    }
    float c = 0;
    for (int t : mesh) {
        const Tri &T = s.tris[t];
        const float va[3] = {(&T.A.x)[a], (&T.B.x)[a], (&T.C.x)[a]};
        for (int k = 0; k < 3; k++) c += va[k];
    }
    c /= mesh.size() * 3;
    std::vector<int> left, right;
    for (int t : mesh) {
        const Tri &T = s.tris[t];
        bool isLeft = (&T.A.x)[a] < c || (&T.B.x)[a] < c || (&T.C.x)[a] < c;
        (isLeft ? left : right).push_back(t);
    }
    int l = buildRef(s, m, left, (a + 1) % 3, depth + 1);
    int r = buildRef(s, m, right, (a + 1) % 3, depth + 1);
    m.nodes[id].left = l;
    m.nodes[id].right = r;
    return id;
}

// Triangle::intersect, MeshLoader.h:70-124
inline bool triIntersect(const Tri &T, V3 o, V3 d, Hit &hit) {
    float ddotN = dot(d, T.n);
    if (ddotN == 0) return false;
    float podotN = dot(T.A - o, T.n);
    float t = podotN / ddotN;
    if (t < 0) return false;
    V3 td = o + t * d;
    V3 w0 = cross(T.B - td, T.C - td);
    V3 w1 = cross(T.C - td, T.A - td);
    V3 w2 = cross(T.A - td, T.B - td);
    float d0 = dot(T.n, w0), d1 = dot(T.n, w1), d2 = dot(T.n, w2);
    if (d0 < 0 || d1 < 0 || d2 < 0) return false;
    hit = Hit();
    if (T.vertexNormals) {
        float a0 = (float)(length(w0) * (d0 >= 0 ? 1 : -1) * 0.5);
        float a1 = (float)(length(w1) * (d1 >= 0 ? 1 : -1) * 0.5);
        float a2 = (float)(length(w2) * (d2 >= 0 ? 1 : -1) * 0.5);
        float totA = a0 + a1 + a2;
        hit.normal = normalize((a0 / totA) * T.nA + (a1 / totA) * T.nB + (a2 / totA) * T.nC);
    } else {
        hit.normal = T.n;
    }
    hit.P = td;
    hit.distance = t;
    hit.hit = true;
    hit.ns = T.n;
    if (T.vertexTextures) {
        float alpha = d0 / length(w0), beta = d1 / length(w1), gamma = d2 / length(w2);
        hit.uv = alpha * T.tA + beta * T.tB + gamma * T.tC;
    }
    return true;
}

// bvh_node::bhv_intersect + the candidate loop of MeshLoader::intersect, MeshLoader.h:265-284,399-404
void walkRef(const Scene &s, const Mesh &m, int node, V3 o, V3 d, Hit &best) {
    const RefNode &n = m.nodes[node];
    if (n.left < 0 && n.right < 0) {
        for (int t : n.tris) {
            Hit h;
            if (triIntersect(s.tris[t], o, d, h) && h.distance < best.distance) {
                best = h;
                best.tri = t;
            }
        }
        return;
    }
    bool lh = m.nodes[n.left].box.hit(o, d), rh = m.nodes[n.right].box.hit(o, d);
    if (lh) walkRef(s, m, n.left, o, d, best);
    if (rh) walkRef(s, m, n.right, o, d, best);
}

// MeshLoader::intersect, MeshLoader.h:392-407
Hit meshIntersect(const Scene &s, const Mesh &m, V3 o, V3 d) {
    Hit best;
    best.hit = false;
    best.distance = INFINITY;
    if (!m.box.hit(o, d)) return best;
    walkRef(s, m, 0, o, d, best);
    best.object = m.object;
    return best;
}

// Sphere::intersect, Objects.h:38-103
Hit sphereIntersect(const Scene &s, const Sphere &sp, V3 o, V3 d) {
    const rt_material &mat = s.materials[s.object_material[sp.object]];
    V3 no = mulPoint(sp.Minv, o, 1.0f);
    V3 nd = normalize(mulPoint(sp.Minv, d, 0.0f));
    const V3 center = v3s(0.0f);
    const float radius = 1.0f;
    V3 c = center - no;
    float cdotc = dot(c, c), cdotd = dot(c, nd);
    Hit hit;
    double D = 0;
    if (cdotc > cdotd * cdotd) D = std::sqrt(cdotc - cdotd * cdotd);  // float sqrt, stored in a double
    if (D <= radius) {
        hit.hit = true;
        float t1 = (float)(cdotd - std::sqrt(radius * radius - D * D));  // double sqrt
        float t2 = (float)(cdotd + std::sqrt(radius * radius - D * D));
        float t = (t1 < 0) ? t2 : t1;
        if (t < 0) {
            hit.hit = false;
            return hit;
        }
        V3 ni = no + t * nd;
        V3 nn = normalize(ni - center);
        hit.P = mulPoint(sp.M, ni, 1.0f);
        hit.distance = length(hit.P - o);
        hit.object = sp.object;
        V3 ng = normalize(mulPoint(sp.N, nn, 0.0f));
        hit.normal = ng;
        hit.ns = ng;
        hit.uv.x = (float)((std::asin((double)nn.y) + M_PI / 2) / M_PI);
        hit.uv.y = (float)((std::atan2((double)nn.z, (double)nn.x) + M_PI) / (2 * M_PI));
        if (mat.has_normal_map) {
            V3 tangent = normalize(cross(v3(0.0f, 1.0f, 0.0f), hit.P));
            V3 bitangent = normalize(cross(hit.normal, tangent));
            hit.T = tangent;
            hit.B = bitangent;
            V3 nm = normalize(evalTexture(s, mat.normal_map, hit.uv));
            // mat3(T,B,N) * nm, glm/detail/type_mat3x3.inl:468-474
            V3 r = v3(tangent.x * nm.x + bitangent.x * nm.y + hit.normal.x * nm.z,
                      tangent.y * nm.x + bitangent.y * nm.y + hit.normal.y * nm.z,
                      tangent.z * nm.x + bitangent.z * nm.y + hit.normal.z * nm.z);
            hit.ns = normalize(r);
        }
    } else {
        hit.hit = false;
    }
    return hit;
}

// Cone::intersect, Objects.h:212-276 (its base is Plane(point (0,1,0), normal (0,1,0)), Objects.h:138-158)
Hit coneIntersect(const Scene &s, const Sphere &cn, V3 o, V3 d) {
    Hit hit;
    const float height = 1.0;
    V3 no = mulPoint(cn.Minv, o, 1.0f);
    V3 nd = normalize(mulPoint(cn.Minv, d, 0.0f));
    float a = nd.x * nd.x + nd.z * nd.z - nd.y * nd.y;
    float b = 2 * (no.x * nd.x + no.z * nd.z - no.y * nd.y);
    float c = no.x * no.x + no.z * no.z - no.y * no.y;
    float delta = b * b - 4 * a * c;
    if (delta < 0) return hit;
    float t1 = (-b - std::sqrt(delta)) / (2 * a);
    float t2 = (-b + std::sqrt(delta)) / (2 * a);
    float t = t1;
    V3 ni = no + t * nd;
    if (t < 0 || ni.y > height || ni.y < 0) {
        t = t2;
        ni = no + t * nd;
        if (t < 0 || ni.y > height || ni.y < 0) return hit;
    }
    V3 nn = normalize(v3(ni.x, -ni.y, ni.z));
    {  // base->intersect(Ray(newOrigin, newDirection)), Objects.h:246-253
        const V3 bn = v3(0.0f, 1.0f, 0.0f);
        float ddotN = dot(nd, bn);
        if (ddotN != 0) {
            float podotN = dot(bn - no, bn);
            float tb = podotN / ddotN;
            if (!(tb < 0)) {
                V3 bi = no + tb * nd;
                if (tb < t && length(bi - bn) <= 1.0) {
                    ni = bi;
                    nn = bn;
                }
            }
        }
    }
    hit.hit = true;
    hit.object = cn.object;
    hit.P = mulPoint(cn.M, ni, 1.0f);
    hit.distance = length(hit.P - o);  // glm::distance
    V3 ng = normalize(mulPoint(cn.N, nn, 0.0f));
    hit.normal = ng;
    hit.ns = ng;  // "cone has no normal map available"
    hit.uv.x = (float)((std::asin((double)nn.y) + M_PI / 2) / M_PI);
    hit.uv.y = (float)((std::atan2((double)nn.z, (double)nn.x) + M_PI) / (2 * M_PI));
    return hit;
}

// Plane::intersect, Objects.h:138-182
Hit planeIntersect(const Scene &s, const Plane &pl, V3 o, V3 d) {
    const rt_material &mat = s.materials[s.object_material[pl.object]];
    Hit hit;
    float ddotN = dot(d, pl.normal);
    if (ddotN == 0) return hit;
    float podotN = dot(pl.point - o, pl.normal);
    float t = podotN / ddotN;
    if (t < 0) return hit;
    hit.P = o + t * d;
    hit.normal = pl.normal;
    hit.distance = t;
    hit.object = pl.object;
    hit.hit = true;
    hit.uv.x = 0.1f * hit.P.x;
    hit.uv.y = 0.1f * hit.P.z;
    hit.ns = pl.normal;
    if (mat.has_normal_map) {
        V3 tangent = v3(0, 0, 1), bitangent = v3(1, 0, 0);
        hit.T = tangent;
        hit.B = bitangent;
        V3 nm = normalize(evalTexture(s, mat.normal_map, hit.uv));
        V3 r = v3(tangent.x * nm.x + bitangent.x * nm.y + hit.normal.x * nm.z,
                  tangent.y * nm.x + bitangent.y * nm.y + hit.normal.y * nm.z,
                  tangent.z * nm.x + bitangent.z * nm.y + hit.normal.z * nm.z);
        hit.ns = normalize(r);
    }
    return hit;
}

Hit objectIntersect(const Scene &s, const Scene::Ref &r, V3 o, V3 d) {
    if (r.kind == 0) return meshIntersect(s, s.meshes[r.idx], o, d);
    if (r.kind == 1) return sphereIntersect(s, s.spheres[r.idx], o, d);
    if (r.kind == 3) return coneIntersect(s, s.cones[r.idx], o, d);
    return planeIntersect(s, s.planes[r.idx], o, d);
}

// closest, main.cpp:59-71
// Test aid (ro_set_ray_log): every closest-hit ray with its result, as raw float bits, one line per ray -- the rays a frame
// really traces (primary rays and every bounce) can then be handed to rt_trace_batch one by one
// (tests/test_gpu_raylog.py, scripts/gpu_r3_pixel_rays.py).
FILE *g_rayLog = nullptr;
FILE *rayLog() { return g_rayLog; }

Hit closest(const Scene &s, V3 o, V3 d, Counters &cn) {
    cn.closest++;
    Hit best;
    best.hit = false;
    best.distance = INFINITY;
    for (const Scene::Ref &r : s.order) {
        Hit h = objectIntersect(s, r, o, d);
        if (h.hit && h.distance < best.distance) best = h;
    }
    if (FILE *f = rayLog()) {
        const float v[7] = {o.x, o.y, o.z, d.x, d.y, d.z, best.distance};
        uint32_t b[7];
        std::memcpy(b, v, sizeof b);
#pragma omp critical(ro_ray_log)
        {
            std::fprintf(f, "%08x %08x %08x %08x %08x %08x %d %d %lld %08x\n", b[0], b[1], b[2], b[3], b[4], b[5], best.hit ? 1 : 0,
                         best.hit ? best.object : -1, best.hit ? (long long)best.tri : -1ll, b[6]);
        }
    }
    return best;
}

// is_shadowed, main.cpp:41-57
bool isShadowed(const Scene &s, V3 point, V3 normal, V3 direction, float distance, Counters &cn) {
    if (dot(normal, direction) < 0) {
        cn.culled++;
        return true;
    }
    cn.shadow++;
    V3 o = point + 0.001f * direction;
    for (const Scene::Ref &r : s.order) {
        Hit h = objectIntersect(s, r, o, direction);
        if (h.hit && h.distance <= distance) return true;
    }
    return false;
}

// pixelAt, Textures.h:156-166.  y == height (fmod(v,1) == 0) reads one row past the image in the reference (UB,
// SURVEY.md Q15); pinned here as "repeat the last row".
V3 pixelAt(const Scene &s, int slot, V2 uv) {
    const int W = s.bw[slot], H = s.bh[slot], C = s.bc[slot];
    if (W == 0) return v3s(0.0f);
    int x = (int)(uint16_t)(int)(W * std::fmod((double)(uv.x * 1.0f), 1.0));
    int y = (int)(uint16_t)(int)(H - H * std::fmod((double)(uv.y * 1.0f), 1.0));
    if (y >= H) y = H - 1;
    if (x >= W) x = W - 1;
    const uint8_t *p = &s.bmp[slot][(size_t)C * ((size_t)W * y + x)];
    return v3(p[2] / 255.f, p[1] / 255.f, p[0] / 255.f);
}

// perlinCalculations, Textures.h:12-19
V3 perlinCalc(V2 uv, V3 us, V3 vs, V3 off) {
    float r = perlin3(uv.x * us.x, uv.y * vs.x, off.x);
    float g = perlin3(uv.x * us.y, uv.y * vs.y, off.y);
    float b = perlin3(uv.x * us.z, uv.y * vs.z, off.z);
    return v3(r, g, b);
}
inline V3 lerpCols(V3 p, V3 c1, V3 c2) {  // Textures.h:94-96 / 109-111
    return v3(p.x * c1.x + ((1.0f - p.x) * c2.x), p.y * c1.y + ((1.0f - p.y) * c2.y), p.z * c1.z + ((1.0f - p.z) * c2.z));
}

V3 evalTexture(const Scene &s, int id, V2 uv) {
    switch (id) {
        case RT_TEX_PERLIN_NORMAL: {  // Textures.h:22-36
            V3 p = perlinCalc(uv, v3(0.0f, 0.0f, 50.f), v3(0.0f, 0.0f, 50.0f), v3(0.0f, 0.0f, 50.0f));
            p = 2.0f * p - v3s(1.0f);
            p = -p;
            p = (p + v3s(1.0f)) / 2.0f;
            return normalize(v3(uv.x, uv.y, p.z));
        }
        case RT_TEX_PERLIN_WATER: {  // Textures.h:39-52
            V3 p = perlinCalc(uv, v3(0.0f, 0.0f, 50.f), v3(0.0f, 0.0f, 50.0f), v3(0.0f, 0.0f, 50.0f));
            p = 2.0f * p - v3s(1.0f);
            p = normalize(p);
            p = (p + v3s(1.0f)) / 2.0f;
            return normalize(v3(uv.x, uv.y, p.z));
        }
        case RT_TEX_OPAL: {  // Textures.h:55-73
            V3 p = perlinCalc(uv, v3(20.0f, 10.f, 15.0f), v3s(20.0f), v3(0.2f, 0.3f, 0.4f));
            p = 2.0f * p - v3s(1.0f);
            p = normalize(p);
            p = (p + v3s(1.0f)) / 2.0f;
            return v3(0.5f + 0.4f * p.x, 0.5f + 0.3f * p.y, 0.5f + 0.2f * p.z);
        }
        case RT_TEX_PERLIN_TERRAIN: {  // Textures.h:76-84
            V3 p = perlinCalc(uv, v3s(30.0f), v3s(30.0f), v3s(0.0f));
            return v3(0.2f + 0.4f * p.x, 0.2f + 0.3f * p.y, 0.2f + 0.2f * p.z);
        }
        case RT_TEX_PERLIN_ICE_TERRAIN: {  // Textures.h:87-99
            V3 p = perlinCalc(uv, v3s(10.0f), v3s(10.0f), v3s(10.0f));
            return lerpCols(p, v3((float)0.722, (float)0.961, (float)0.937), v3((float)0.075, (float)0.482, (float)0.631));
        }
        case RT_TEX_SNOW_TERRAIN: {  // Textures.h:102-114
            V3 p = perlinCalc(uv, v3s(10.0f), v3s(10.0f), v3s(10.0f));
            return lerpCols(p, v3((float)0.722, (float)0.961, (float)0.937), v3((float)0.212, (float)0.51, (float)0.62));
        }
        case RT_TEX_QWILFISH: {  // Textures.h:123-132
            float y = 0.5f * (uv.y + 3.0f);
            if (y < 1) return v3((float)0.937, (float)0.922, (float)0.392);
            return v3(0.0f, (float)0.416, (float)0.42);
        }
        case RT_TEX_BMP_COLOR: return pixelAt(s, 0, uv);      // Textures.h:173
        case RT_TEX_BMP_NORMAL: return pixelAt(s, 1, uv);     // :178
        case RT_TEX_BMP_ROUGHNESS: { V3 c = pixelAt(s, 2, uv); return v3s(c.x); }  // :186
        case RT_TEX_BMP_OCCLUSION: { V3 c = pixelAt(s, 3, uv); return v3s(c.x); }  // :182
    }
    return v3s(0.0f);
}

// glm::reflect / glm::refract, glm/detail/func_geometric.inl:104-123
inline V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0f; }
inline V3 refract(V3 I, V3 N, float eta) {
    float dv = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - dv * dv);
    if (k >= 0.0f) return eta * I - (eta * dv + std::sqrt(k)) * N;
    return v3s(0.0f);
}
inline float angle(V3 x, V3 y) { return std::acos(gclamp(dot(x, y), -1.0f, 1.0f)); }  // glm/gtx/vector_angle.inl:16-21

// PhongModel, main.cpp:81-213
V3 phong(const Scene &s, V3 point, V3 normal, V3 ns, V2 uv, V3 view, const rt_material &mat, int maxBounces,
         const Hit &hit, Counters &cn) {
    cn.shaded++;
    V3 color = v3s(0.0f);
    if (dot(normal, view) < 0) normal = -normal;
    if (dot(ns, view) < 0) ns = -ns;
    const V3 mdiff = v3(mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]);
    const V3 mspec = v3(mat.specular[0], mat.specular[1], mat.specular[2]);
    for (const Light &light : s.lights) {
        V3 ld = normalize(light.pos - point);
        const float dl = length(light.pos - point);
        if (!isShadowed(s, point, normal, ld, dl, cn)) {
            V3 dcol = mat.texture != RT_TEX_NONE ? evalTexture(s, mat.texture, uv) : mdiff;
            const float diffuse = smax(0.0f, dot(ld, ns));
            V3 h = normalize(ld + view);
            const float distance = smax(0.1f, dl);
            const float attenuation = (float)(1.0f / std::pow((double)distance, 2.0));
            V3 diffusion = attenuation * light.color * dcol * diffuse;
            V3 specular_term = v3s(0.0f);
            float shiny;
            if (mat.has_img_texture) {
                float rough = evalTexture(s, mat.roughness, uv).x;
                shiny = (float)((0.5f / std::pow((double)rough, 4.0)) - 0.5f);
            } else {
                shiny = mat.shininess;
            }
            if (mat.is_anisotropic) {
                float NdotL = dot(ns, ld), NdotV = dot(ns, view);
                if (NdotL > 0 && NdotV > 0) {
                    float HdotT = dot(h, hit.T), HdotB = dot(h, hit.B), HdotN = dot(h, ns);
                    float exponent = -2.0f * (std::pow(HdotT / mat.alpha_x, 2.0f) * std::pow(HdotB / mat.alpha_y, 2.0f)) / (1 + HdotN);
                    const float pi = 3.14159265358979323846264338327950288f;
                    specular_term = (mspec * NdotL * std::exp(exponent)) / (std::sqrt(NdotL * NdotV) * 4 * pi * mat.alpha_x * mat.alpha_y);
                }
            } else {
                const float specular = smax(0.0f, std::pow(dot(h, ns), 4 * shiny));
                specular_term = attenuation * light.color * mspec * specular;
            }
            color = color + (diffusion + specular_term);
        }
    }
    if (maxBounces > 0) {
        V3 reflection = v3s(0.0f);
        if (mat.reflection > 0) {
            color = color * (1 - mat.reflection);
            V3 rd = reflect(-view, ns);
            V3 rp = point + 0.001f * rd;
            Hit ch = closest(s, rp, rd, cn);
            if (ch.hit)
                reflection = mat.reflection * phong(s, ch.P, ch.normal, ch.ns, ch.uv, normalize(-rd),
                                                    s.materials[s.object_material[ch.object]], maxBounces - 1, ch, cn);
        }
        V3 refraction = v3s(0.0f);
        if (mat.refraction > 0) {
            color = color * (1 - mat.refraction);
            const bool entering = dot(ns, -view) < 0.0f;
            const float n1 = entering ? 1.0f : mat.sigma, n2 = entering ? mat.sigma : 1.0f;
            const float eta = n1 / n2;
            V3 rd = refract(-view, entering ? ns : -ns, eta);
            V3 rp = point + 0.001f * rd;
            Hit ch = closest(s, rp, rd, cn);
            if (ch.hit) {
                refraction = mat.refraction * phong(s, ch.P, ch.normal, ch.ns, ch.uv, normalize(-rd),
                                                    s.materials[s.object_material[ch.object]], maxBounces - 1, ch, cn);
                float O1 = std::cos(angle(ns, view));
                float O2 = std::cos(angle(-ns, rd));
                float R = (float)(0.5f * (std::pow((double)((n1 * O1 - n2 * O2) / (n1 * O1 + n2 * O2)), 2.0) +
                                          std::pow((double)((n1 * O2 - n2 * O1) / (n1 * O2 + n2 * O1)), 2.0)));
                float T = 1 - R;
                reflection = reflection * R;
                refraction = refraction * T;
            }
        }
        color = color + (reflection + refraction);
    }
    if (mat.has_img_texture) {
        color = color + s.ambient * 0.1f * evalTexture(s, mat.occlusion, uv).x;
    } else {
        color = color + s.ambient * v3(mat.ambient[0], mat.ambient[1], mat.ambient[2]);
    }
    return color;
}

// trace_ray, main.cpp:221-233
V3 traceRay(const Scene &s, V3 o, V3 d, int bounces, Counters &cn, Hit *out_hit = nullptr) {
    Hit ch = closest(s, o, d, cn);
    if (out_hit) *out_hit = ch;
    V3 color = v3s(0.0f);
    if (ch.hit)
        color = phong(s, ch.P, ch.normal, ch.ns, ch.uv, normalize(-d), s.materials[s.object_material[ch.object]], bounces, ch, cn);
    return v3(gclamp(color.x, 0.0f, 1.0f), gclamp(color.y, 0.0f, 1.0f), gclamp(color.z, 0.0f, 1.0f));
}

// toneMapping, main.cpp:240-256
V3 toneMap(V3 in) {
    const float alpha = 1.5f, beta = 1.8f, gamma = 2.2f, oog = 1.0f / gamma;
    V3 n = v3(std::pow(alpha * std::pow(in.x, beta), oog), std::pow(alpha * std::pow(in.y, beta), oog),
              std::pow(alpha * std::pow(in.z, beta), oog));
    V3 t = v3(gmin(n.x, 1.0f), gmin(n.y, 1.0f), gmin(n.z, 1.0f));
    return v3(gclamp(t.x, 0.0f, 1.0f), gclamp(t.y, 0.0f, 1.0f), gclamp(t.z, 0.0f, 1.0f));
}

// shared counter-based jitter; same formula in oracle/ref_harness.cpp and csrc/rt_device.h
inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
inline void hashJitter(uint32_t pixel, uint32_t sample, uint32_t seed, float &jx, float &jy) {
    uint32_t a = hash32(pixel ^ (seed * 0x85ebca6bU + 0x9e3779b9U));
    uint32_t b = hash32(a + 2u * sample + 1u), c = hash32(a + 2u * sample + 2u);
    jx = ((float)(b >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1.5f;
    jy = ((float)(c >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1.5f;
}
const float kJitter4[8] = {-1.0 / 4.0, 3.0 / 4.0, 3.0 / 4.0, 1.0 / 3.0, -3.0 / 4.0, -1.0 / 4.0, 1.0 / 4.0, -3.0 / 4.0};  // main.cpp:601-606

uint32_t primIdOf(const Hit &h) {
    if (!h.hit) return RT_PRIM_MISS;
    if (h.tri >= 0) return (uint32_t)h.tri;
    return RT_PRIM_ANALYTIC | (uint32_t)h.object;
}

void fillHit(const Scene &s, const Hit &h, rt_hit &o) {
    std::memset(&o, 0, sizeof o);
    o.hit = h.hit ? 1 : 0;
    o.object = h.hit ? h.object : -1;
    o.triangle = -1;
    if (h.hit && h.tri >= 0)
        for (const Mesh &m : s.meshes)
            if (m.object == h.object) o.triangle = (int32_t)(h.tri - m.first);
    o.prim = primIdOf(h);
    o.distance = h.distance;
    if (!h.hit) return;
    const V3 *src[5] = {&h.P, &h.normal, &h.ns, &h.T, &h.B};
    float *dst[5] = {o.intersection, o.normal, o.normal_shading, o.tangent, o.bitangent};
    for (int k = 0; k < 5; k++) {
        dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z;
    }
    o.uv[0] = h.uv.x;
    o.uv[1] = h.uv.y;
}

M4 loadM4(const float *p) {
    M4 m;
    std::memcpy(m.c, p, 64);
    return m;
}

}  // namespace

struct ro_scene {
    Scene s;
};

extern "C" {

ro_scene *ro_create(const ro_scene_desc *d) {
    ro_scene *rs = new ro_scene();
    Scene &s = rs->s;
    s.materials.assign(d->materials, d->materials + d->n_materials);
    s.object_material.assign(d->object_material, d->object_material + d->n_objects);
    s.tris.resize(d->n_triangles);
    auto g3 = [](const float *p, int64_t k) { return v3(p[3 * k], p[3 * k + 1], p[3 * k + 2]); };
    auto g2 = [](const float *p, int64_t k) { return V2{p[2 * k], p[2 * k + 1]}; };
    for (int64_t k = 0; k < d->n_triangles; k++) {
        Tri &T = s.tris[k];
        T.A = g3(d->v0, k); T.B = g3(d->v1, k); T.C = g3(d->v2, k);
        T.n = normalize(cross(T.B - T.A, T.C - T.A));  // Triangle ctor, MeshLoader.h:58
        T.nA = g3(d->nA, k); T.nB = g3(d->nB, k); T.nC = g3(d->nC, k);
        T.tA = g2(d->tA, k); T.tB = g2(d->tB, k); T.tC = g2(d->tC, k);
        T.vertexNormals = d->tri_flags[k] & 1;
        T.vertexTextures = d->tri_flags[k] & 2;
    }
    for (int k = 0; k < d->n_spheres; k++)
        s.spheres.push_back(Sphere{loadM4(d->sph_M + 16 * k), loadM4(d->sph_Minv + 16 * k), loadM4(d->sph_N + 16 * k), d->sph_object[k]});
    for (int k = 0; k < d->n_cones; k++)
        s.cones.push_back(Sphere{loadM4(d->cone_M + 16 * k), loadM4(d->cone_Minv + 16 * k), loadM4(d->cone_N + 16 * k), d->cone_object[k]});
    for (int k = 0; k < d->n_planes; k++) s.planes.push_back(Plane{g3(d->pl_point, k), g3(d->pl_normal, k), d->pl_object[k]});
    for (int k = 0; k < d->n_lights; k++) s.lights.push_back(Light{g3(d->light_pos, k), g3(d->light_color, k)});
    s.ambient = v3(d->ambient[0], d->ambient[1], d->ambient[2]);
    for (int k = 0; k < 4; k++) {
        s.bw[k] = d->bmp_w[k]; s.bh[k] = d->bmp_h[k]; s.bc[k] = d->bmp_c[k];
        if (d->bmp_w[k] > 0) s.bmp[k].assign(d->bmp[k], d->bmp[k] + (size_t)d->bmp_w[k] * d->bmp_h[k] * d->bmp_c[k]);
    }
    // meshes: maximal runs of equal tri_object
    int64_t k = 0;
    while (k < d->n_triangles) {
        int64_t e = k;
        while (e < d->n_triangles && d->tri_object[e] == d->tri_object[k]) e++;
        Mesh m;
        m.object = d->tri_object[k];
        m.first = k;
        m.count = e - k;
        std::vector<int> all;
        for (int64_t t = k; t < e; t++) {
            all.push_back((int)t);
            m.box.grow(s.tris[t].A); m.box.grow(s.tris[t].B); m.box.grow(s.tris[t].C);
        }
        buildRef(s, m, all, 0, 0);
        s.meshes.push_back(std::move(m));
        k = e;
    }
    s.order.assign(d->n_objects, Scene::Ref{-1, -1});
    for (size_t i = 0; i < s.meshes.size(); i++) s.order[s.meshes[i].object] = Scene::Ref{0, (int)i};
    for (size_t i = 0; i < s.spheres.size(); i++) s.order[s.spheres[i].object] = Scene::Ref{1, (int)i};
    for (size_t i = 0; i < s.planes.size(); i++) s.order[s.planes[i].object] = Scene::Ref{2, (int)i};
    for (size_t i = 0; i < s.cones.size(); i++) s.order[s.cones[i].object] = Scene::Ref{3, (int)i};
    std::vector<Scene::Ref> live;
    for (auto &r : s.order)
        if (r.kind >= 0) live.push_back(r);
    s.order = live;
    return rs;
}

void ro_destroy(ro_scene *s) { delete s; }

// path: file to (over)write with one line per closest-hit ray from now on; NULL or "" stops logging
int ro_set_ray_log(const char *path) {
    if (g_rayLog) std::fclose(g_rayLog);
    g_rayLog = nullptr;
    if (path && *path) {
        g_rayLog = std::fopen(path, "w");
        if (!g_rayLog) return 1;
    }
    return 0;
}

int ro_render(const ro_scene *rs, const rt_camera *cam, const rt_render_params *p, uint8_t *out_rgb8,
              int32_t *out_rgb8_i32, float *out_rgbf, uint32_t *out_primid, rt_stats *stats, int threads) {
    const Scene &s = rs->s;
    const int width = p->width, height = p->height, spp = p->spp;
    const bool table = p->jitter_mode == RT_JITTER_TABLE || (p->jitter_mode == RT_JITTER_AUTO && spp == 4);
    if (table && spp != 4) return 1;
    // main.cpp:561-563
    const float fov = cam->fov_deg;
    const float sc = 2 * tan(0.5 * fov / 180 * M_PI) / width;
    const float X = -sc * width / 2;
    const float Y = sc * height / 2;
    M4 R = loadM4(cam->rot);
    const V3 origin = v3(cam->origin[0], cam->origin[1], cam->origin[2]);
    const int tile_size = 16;
    const int tiles_x = (width + tile_size - 1) / tile_size, tiles_y = (height + tile_size - 1) / tile_size;
    const int world = p->tile_world > 0 ? p->tile_world : 1;
    if (threads <= 0) threads = omp_get_max_threads();
    int only_i = -1, only_j = -1;  // RO_ONLY_PIXEL="i,j": every other pixel stays black (debugging aid, with ro_set_ray_log)
    if (const char *e = std::getenv("RO_ONLY_PIXEL"))
        if (std::sscanf(e, "%d,%d", &only_i, &only_j) != 2) only_i = only_j = -1;
    Counters total;
    double t0 = omp_get_wtime();
#pragma omp parallel num_threads(threads)
    {
        Counters cn;
#pragma omp for schedule(dynamic, 1)
        for (int tile = 0; tile < tiles_x * tiles_y; tile++) {
            if (tile % world != p->tile_rank) continue;
            const int tj = tile / tiles_x, ti = tile - tj * tiles_x;
            for (int i = ti * tile_size; i < std::min(ti * tile_size + tile_size, width); i++)
                for (int j = tj * tile_size; j < std::min(tj * tile_size + tile_size, height); j++) {
                    if (only_i >= 0 && (i != only_i || j != only_j)) continue;  // (debugging aid, below)
                    V3 pixel = v3s(0.0f);
                    uint32_t pid = RT_PRIM_MISS;
                    for (int sample = 0; sample < spp; ++sample) {
                        float jx, jy;
                        if (table) {
                            jx = kJitter4[2 * sample];
                            jy = kJitter4[2 * sample + 1];
                        } else {
                            hashJitter((uint32_t)(j * width + i), (uint32_t)sample, p->seed, jx, jy);
                        }
                        float dx = X + (i + jx) * sc + sc / 2;  // main.cpp:629-631
                        float dy = Y - (j + jy) * sc - sc / 2;
                        V3 d = normalize(mulPoint(R, v3(dx, dy, cam->dz), 0.0f));
                        Hit h;
                        V3 c = traceRay(s, origin, d, p->max_depth, cn, sample == 0 ? &h : nullptr);
                        if (sample == 0) pid = primIdOf(h);
                        pixel = pixel + c;
                    }
                    pixel = pixel / (float)spp;  // main.cpp:646
                    const size_t q = (size_t)j * width + i;
                    if (out_rgbf) {
                        out_rgbf[3 * q] = pixel.x; out_rgbf[3 * q + 1] = pixel.y; out_rgbf[3 * q + 2] = pixel.z;
                    }
                    V3 tm = toneMap(pixel);
                    const int r = (int)(float)(255 * tm.x), g = (int)(float)(255 * tm.y), b = (int)(float)(255 * tm.z);  // Image.h:91-95
                    if (out_rgb8_i32) {
                        out_rgb8_i32[3 * q] = r; out_rgb8_i32[3 * q + 1] = g; out_rgb8_i32[3 * q + 2] = b;
                    }
                    if (out_rgb8) {
                        out_rgb8[3 * q] = (uint8_t)r; out_rgb8[3 * q + 1] = (uint8_t)g; out_rgb8[3 * q + 2] = (uint8_t)b;
                    }
                    if (out_primid) out_primid[q] = pid;
                }
        }
#pragma omp critical
        {
            total.closest += cn.closest; total.shadow += cn.shadow; total.culled += cn.culled; total.shaded += cn.shaded;
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->rays_closest = total.closest;
        stats->rays_shadow = total.shadow;
        stats->shadow_culled = total.culled;
        stats->hits_shaded = total.shaded;
        stats->ms_total = (omp_get_wtime() - t0) * 1e3;
        stats->n_triangles = s.tris.size();
        int64_t ntile = 0;
        for (int tile = 0; tile < tiles_x * tiles_y; tile++) ntile += (tile % world == p->tile_rank);
        stats->primary_samples = 0;
        for (int tile = 0; tile < tiles_x * tiles_y; tile++) {
            if (tile % world != p->tile_rank) continue;
            const int tj = tile / tiles_x, ti = tile - tj * tiles_x;
            const int w = std::min(ti * tile_size + tile_size, width) - ti * tile_size;
            const int h = std::min(tj * tile_size + tile_size, height) - tj * tile_size;
            stats->primary_samples += (uint64_t)w * h * spp;
        }
    }
    return 0;
}

int ro_trace_batch(const ro_scene *rs, int64_t n, const float *rays, int32_t depth, float *out_color,
                   rt_hit *out_hits, uint8_t *out_shadow, rt_stats *stats, int threads) {
    const Scene &s = rs->s;
    if (threads <= 0) threads = omp_get_max_threads();
    Counters total;
#pragma omp parallel num_threads(threads)
    {
        Counters cn;
#pragma omp for schedule(dynamic, 16)
        for (int64_t k = 0; k < n; k++) {
            V3 o = v3(rays[6 * k], rays[6 * k + 1], rays[6 * k + 2]), d = v3(rays[6 * k + 3], rays[6 * k + 4], rays[6 * k + 5]);
            Hit h;
            V3 c = traceRay(s, o, d, depth, cn, &h);
            if (out_color) {
                out_color[3 * k] = c.x; out_color[3 * k + 1] = c.y; out_color[3 * k + 2] = c.z;
            }
            if (out_hits) fillHit(s, h, out_hits[k]);
            if (out_shadow) {
                for (int l = 0; l < 4; l++) out_shadow[4 * k + l] = 0;
                if (h.hit) {
                    Counters scratch;
                    V3 view = normalize(-d), nrm = h.normal;
                    if (dot(nrm, view) < 0) nrm = -nrm;
                    for (size_t l = 0; l < s.lights.size() && l < 4; l++) {
                        V3 ld = normalize(s.lights[l].pos - h.P);
                        float dl = length(s.lights[l].pos - h.P);
                        out_shadow[4 * k + l] = isShadowed(s, h.P, nrm, ld, dl, scratch) ? 1 : 0;
                    }
                }
            }
        }
#pragma omp critical
        {
            total.closest += cn.closest; total.shadow += cn.shadow; total.culled += cn.culled; total.shaded += cn.shaded;
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->primary_samples = (uint64_t)n;
        stats->rays_closest = total.closest;
        stats->rays_shadow = total.shadow;
        stats->shadow_culled = total.culled;
        stats->hits_shaded = total.shaded;
    }
    return 0;
}

int ro_eval_texture(const ro_scene *rs, int32_t id, int64_t n, const float *uv, float *out) {
    for (int64_t k = 0; k < n; k++) {
        V3 c = evalTexture(rs->s, id, V2{uv[2 * k], uv[2 * k + 1]});
        out[3 * k] = c.x; out[3 * k + 1] = c.y; out[3 * k + 2] = c.z;
    }
    return 0;
}

int ro_eval_perlin(int64_t n, const float *p, float *out) {
    for (int64_t k = 0; k < n; k++) out[k] = perlin3(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
    return 0;
}

static void orderRef(const Mesh &m, int node, uint32_t &next, uint32_t *rank) {
    const RefNode &n = m.nodes[node];
    if (n.left < 0 && n.right < 0) {
        for (int t : n.tris) rank[t] = next++;
        return;
    }
    orderRef(m, n.left, next, rank);
    orderRef(m, n.right, next, rank);
}

// The KERNEL's closest-hit walk restated on the CPU, for the counter identity of SURVEY.md 8(d): the committed tree as
// rt_get_bvh hands it over (64-byte 4-wide nodes with 8-bit child boxes, rt_types.h: RtNode4; leaf reference =
// 0x80000000 | (count - 1) << 28 | (n_nodes + first slot); tri_order[slot] = global triangle) is walked for one ray at a
// time with the traversal kernel's own arithmetic and order (raytracer_amd/csrc/rt_kernels.hip: rtInitWalk, rtNodeStep,
// rtTriStep, rtPostponeLeaf and the phase rule of traceQueue for a wave whose lanes all hold the same ray: a triangle step
// whenever a leaf is pending -- node phase iff nNode >= 2 nTri, RT_VOTE_NODE 1/2 -- else a node step), counting node steps and triangle tests.  No analytic
// primitives (the test scene has none).  out_slot = BVH slot of the closest triangle or -1.
int ro_walk_counts(const ro_scene *rs, uint32_t root_ref, float bound, int64_t n_nodes, const uint32_t *nodes, const uint32_t *tri_order,
                   const int32_t *tri_object, const uint32_t *rank, int64_t n_rays, const float *rays, uint64_t *out_nodes,
                   uint64_t *out_tris, int64_t *out_slot, uint32_t *slot_tests /* per BVH slot: tests over all rays, or NULL */) {
    const Scene &s = rs->s;
    const uint32_t LEAF = 0x80000000u, DONE = 0xFFFFFFFFu, MASK = 0x0FFFFFFFu, MISS = 0xFFFFFFFFu;
    auto safeInv = [](float d) {
        float inv = 1.0f / d;
        const float cap = 1e30f;
        if (!(std::fabs(inv) <= cap)) inv = std::signbit(d) ? -cap : cap;
        return inv;
    };
    auto asF = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    auto asU = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto ub = [](uint32_t v, int c) { return (float)((v >> (8 * c)) & 0xFFu); };
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n_rays; i++) {
        const V3 o = v3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), d = v3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        const V3 inv = v3(safeInv(d.x), safeInv(d.y), safeInv(d.z));
        const V3 oi = v3(o.x * inv.x, o.y * inv.y, o.z * inv.z);
        const float u = 4.76837158e-07f;
        const V3 err = v3((std::fabs(oi.x) + bound * std::fabs(inv.x)) * u, (std::fabs(oi.y) + bound * std::fabs(inv.y)) * u,
                          (std::fabs(oi.z) + bound * std::fabs(inv.z)) * u);
        const V3 oiN = oi + err, oiF = oi - err;
        std::vector<uint32_t> stack;
        uint32_t cur = s.tris.empty() ? DONE : root_ref, pend = DONE;
        // (rtFarLimit, rt_kernels.hip: a ray without an analytic hit starts with the farthest distance the scene allows)
        const float reach = 2.0f * (std::fabs(o.x) + std::fabs(o.y) + std::fabs(o.z)) + 4.0f * bound;
        float best_t = reach * std::fmin(std::fmin(std::fabs(inv.x), std::fabs(inv.y)), std::fabs(inv.z));
        uint32_t best_id = MISS, best_rank = 0;
        int best_obj = 0x7FFFFFFF;
        uint64_t nn = 0, nt = 0;
        for (;;) {
            if (pend == DONE && (cur & LEAF) && cur != DONE) {  // rtPostponeLeaf
                pend = cur;
                if (stack.empty()) cur = DONE; else { cur = stack.back(); stack.pop_back(); }
            }
            // (the wave's vote with every lane on the same ray: RT_VOTE_NODE 1/2 -- a pending leaf goes first)
            if (pend == DONE && !(cur & LEAF)) {  // node step
                nn++;
                const uint32_t *q = nodes + 16 * (size_t)(cur & MASK);
                const float org[3] = {asF(q[0]), asF(q[1]), asF(q[2])};
                const float mx = asF(q[3]) * inv.x, my = asF(q[14]) * inv.y, mz = asF(q[15]) * inv.z;
                const float anx = std::fmaf(org[0], inv.x, -oiN.x), any_ = std::fmaf(org[1], inv.y, -oiN.y), anz = std::fmaf(org[2], inv.z, -oiN.z);
                const float afx = std::fmaf(org[0], inv.x, -oiF.x), afy = std::fmaf(org[1], inv.y, -oiF.y), afz = std::fmaf(org[2], inv.z, -oiF.z);
                const bool negx = inv.x < 0.0f, negy = inv.y < 0.0f, negz = inv.z < 0.0f;
                const uint32_t lox = q[8], loy = q[9], loz = q[10], hix = q[11], hiy = q[12], hiz = q[13];
                const uint32_t nrx = negx ? hix : lox, frx = negx ? lox : hix;
                const uint32_t nry = negy ? hiy : loy, fry = negy ? loy : hiy;
                const uint32_t nrz = negz ? hiz : loz, frz = negz ? loz : hiz;
                uint32_t ref[4] = {q[4], q[5], q[6], q[7]}, key[4];
                for (int c = 0; c < 4; c++) {
                    const float tnx = std::fmaf(ub(nrx, c), mx, anx), tfx = std::fmaf(ub(frx, c), mx, afx);
                    const float tny = std::fmaf(ub(nry, c), my, any_), tfy = std::fmaf(ub(fry, c), my, afy);
                    const float tnz = std::fmaf(ub(nrz, c), mz, anz), tfz = std::fmaf(ub(frz, c), mz, afz);
                    const float tn = std::fmax(std::fmax(tnx, tny), std::fmax(tnz, 0.0f));
                    const float tf = std::fmin(std::fmin(tfx, tfy), std::fmin(tfz, best_t)) * 1.000001f;
                    const bool hit = tn <= tf && ref[c] != 0xFFFFFFFFu;
                    key[c] = hit ? asU(tn) : 0xFFFFFFFFu;
                }
                auto cswap = [&](int a, int b) {
                    if (key[b] < key[a]) { std::swap(key[a], key[b]); std::swap(ref[a], ref[b]); }
                };
                cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
                if (key[0] == 0xFFFFFFFFu) {
                    if (stack.empty()) cur = DONE; else { cur = stack.back(); stack.pop_back(); }
                } else {
                    if (key[3] != 0xFFFFFFFFu) stack.push_back(ref[3]);
                    if (key[2] != 0xFFFFFFFFu) stack.push_back(ref[2]);
                    if (key[1] != 0xFFFFFFFFu) stack.push_back(ref[1]);
                    cur = ref[0];
                }
            } else if (pend != DONE) {  // triangle step (rtTriStep / rtTriHit)
                nt++;
                const uint32_t slot = (pend & MASK) - (uint32_t)n_nodes, g = tri_order[slot];
                if (slot_tests) {
#pragma omp atomic
                    slot_tests[slot]++;
                }
                const Tri &T = s.tris[g];
                bool hit = false;
                float t = 0.0f;
                const float ddotN = dot(d, T.n);
                if (ddotN != 0) {
                    const float podotN = dot(T.A - o, T.n);
                    t = podotN / ddotN;
                    if (!(t < 0) && t <= best_t) {
                        const V3 td = o + t * d;
                        const V3 w0 = cross(T.B - td, T.C - td), w1 = cross(T.C - td, T.A - td), w2 = cross(T.A - td, T.B - td);
                        hit = !(dot(T.n, w0) < 0 || dot(T.n, w1) < 0 || dot(T.n, w2) < 0);
                    }
                }
                const int obj = tri_object[g];
                const uint32_t rk = rank[g];
                const bool take = hit && (t < best_t || (t == best_t && best_id != MISS && (obj < best_obj || (obj == best_obj && rk < best_rank))));
                if (take) { best_t = t; best_id = slot; best_obj = obj; best_rank = rk; }
                if ((pend >> 28) & 7u) pend += 1u - 0x10000000u; else pend = DONE;
            }
            if (cur == DONE && pend == DONE) break;
        }
        out_nodes[i] = nn;
        out_tris[i] = nt;
        if (out_slot) out_slot[i] = best_id == MISS ? -1 : (int64_t)best_id;
    }
    return 0;
}

int ro_reference_order(const ro_scene *rs, uint32_t *rank) {
    uint32_t next = 0;
    for (const Mesh &m : rs->s.meshes) orderRef(m, 0, next, rank);
    return 0;
}

}  // extern "C"
