// MeshLoader.h -- Wavefront .obj ingest for the scene-definition surface.
//
// Same constructor as the reference (MeshLoader.h:300): MeshLoader(filename, translation, hasMaterial, material).
// The parser reproduces the reference's line dispatch and its quirks because they decide what gets rendered
// (SURVEY.md App. A Q11-Q13):
//   * dispatch on line[1]=='n' (vn), then line[1]=='t' (vt), then line[0] in {v, s, f}   (MeshLoader.h:323-344)
//   * indices are scanned with %f and converted by `index - 1`                            (:356-357,364-369)
//   * the face format is chosen by the current `s` state and by "has any vt been seen"    (:347,363)
//   * the scan targets persist across lines, so a face line that the chosen format cannot fully parse keeps
//     stale 2nd/3rd indices (64 faces of crystalpillar.obj)                                (:315)
//   * the translation is added to vertices AND to vn normals AND to vt coordinates       (:326,329,334)
// What differs: triangles are plain records in a vector (no per-triangle Object, no per-mesh BVH) -- the
// acceleration structure is built once over the whole flattened scene by rt_commit().
// A missing file is a hard error here (the reference prints and carries on into UB).
#pragma once
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "Object.h"

struct Triangle {
    glm::vec3 vertexA, vertexB, vertexC;
    glm::vec3 normalA, normalB, normalC;
    glm::vec2 textureA, textureB, textureC;
    bool vertexNormals = false;   // any vn != 0          (Triangle ctor, MeshLoader.h:62-64)
    bool vertexTextures = false;  // any vt != 0          (:66-68)
};

class MeshLoader : public Object {
    std::vector<glm::vec3> vertices;
    std::vector<glm::vec3> normals;
    std::vector<glm::vec2> textureCoords;
    std::vector<Triangle> triangles;
    glm::vec3 minBounds, maxBounds;  // over all `v` lines (MeshLoader.h:336-341)

    template <class V>
    static const V &at(const std::vector<V> &v, float one_based, const char *what, const std::string &file) {
        // the reference subscripts with (float index - 1) unchecked; out of range is UB there, an error here
        const float f = one_based - 1;
        if (!(f >= 0) || !(f < (float)v.size())) throw std::runtime_error("MeshLoader: bad " + std::string(what) + " index in " + file);
        return v[(size_t)f];
    }
    void addTriangle(glm::vec3 a, glm::vec3 b, glm::vec3 c, glm::vec3 na = glm::vec3(0.0f),
                     glm::vec3 nb = glm::vec3(0.0f), glm::vec3 nc = glm::vec3(0.0f), glm::vec2 ta = glm::vec2(0.0f),
                     glm::vec2 tb = glm::vec2(0.0f), glm::vec2 tc = glm::vec2(0.0f)) {
        Triangle t;
        t.vertexA = a; t.vertexB = b; t.vertexC = c;
        t.normalA = na; t.normalB = nb; t.normalC = nc;
        t.textureA = ta; t.textureB = tb; t.textureC = tc;
        const glm::vec3 z3(0.0f);
        const glm::vec2 z2(0.0f);
        t.vertexNormals = (na != z3) || (nb != z3) || (nc != z3);
        t.vertexTextures = (ta != z2) || (tb != z2) || (tc != z2);
        triangles.push_back(t);
    }

public:
    MeshLoader(const std::string &filename, glm::vec3 translation, bool hasMaterial, Material m = Material()) {
        if (hasMaterial) setMaterial(m);
        std::ifstream file(filename);
        if (!file.is_open()) throw std::runtime_error("MeshLoader: could not open " + filename);
        minBounds = glm::vec3(INFINITY);
        maxBounds = glm::vec3(-INFINITY);
        // scan targets live across lines on purpose (stale-index quirk)
        float x = 0, y = 0, z = 0, nx = 0, ny = 0, nz = 0, tx = 0, ty = 0, tz = 0;
        int smoothShading = 0;
        bool sawTexture = false;
        std::string line;
        while (std::getline(file, line)) {
            const char c0 = line.size() > 0 ? line[0] : '\0';
            const char c1 = line.size() > 1 ? line[1] : '\0';
            const char *s = line.c_str();
            if (c1 == 'n') {
                sscanf(s, "vn %f %f %f", &x, &y, &z);
                normals.emplace_back(x + translation.x, y + translation.y, z + translation.z);
            } else if (c1 == 't') {
                sscanf(s, "vt %f %f", &x, &y);
                textureCoords.emplace_back(x + translation.x, y + translation.y);
                sawTexture = true;
            } else if (c0 == 'v') {
                sscanf(s, "v %f %f %f", &x, &y, &z);
                glm::vec3 p(x + translation.x, y + translation.y, z + translation.z);
                vertices.push_back(p);
                for (int k = 0; k < 3; k++) {
                    minBounds[k] = std::fmin(minBounds[k], p[k]);
                    maxBounds[k] = std::fmax(maxBounds[k], p[k]);
                }
            } else if (c0 == 's') {
                sscanf(s, "s %d", &smoothShading);
            } else if (c0 == 'f') {
                if (smoothShading == 0) {
                    sscanf(s, "f %f %f %f", &x, &y, &z);
                    addTriangle(at(vertices, x, "vertex", filename), at(vertices, y, "vertex", filename),
                                at(vertices, z, "vertex", filename));
                } else if (sawTexture) {
                    sscanf(s, "f %f/%f/%f %f/%f/%f %f/%f/%f", &x, &tx, &nx, &y, &ty, &ny, &z, &tz, &nz);
                    addTriangle(at(vertices, x, "vertex", filename), at(vertices, y, "vertex", filename),
                                at(vertices, z, "vertex", filename), at(normals, nx, "normal", filename),
                                at(normals, ny, "normal", filename), at(normals, nz, "normal", filename),
                                at(textureCoords, tx, "uv", filename), at(textureCoords, ty, "uv", filename),
                                at(textureCoords, tz, "uv", filename));
                } else {
                    sscanf(s, "f %f//%f %f//%f %f//%f", &x, &nx, &y, &ny, &z, &nz);
                    addTriangle(at(vertices, x, "vertex", filename), at(vertices, y, "vertex", filename),
                                at(vertices, z, "vertex", filename), at(normals, nx, "normal", filename),
                                at(normals, ny, "normal", filename), at(normals, nz, "normal", filename));
                }
            }
        }
    }

    const std::vector<Triangle> &getTriangles() const { return triangles; }
    size_t vertexCount() const { return vertices.size(); }
    size_t normalCount() const { return normals.size(); }
    size_t uvCount() const { return textureCoords.size(); }
    glm::vec3 getMinBounds() const { return minBounds; }
    glm::vec3 getMaxBounds() const { return maxBounds; }
    // replicate the parsed mesh at an extra offset (used by the synthetic stress scene only)
    void translateAll(glm::vec3 d) {
        for (Triangle &t : triangles) {
            t.vertexA = t.vertexA + d; t.vertexB = t.vertexB + d; t.vertexC = t.vertexC + d;
        }
    }
    void flatten(FlatScene &out, int object_id) const override;
    uint64_t geometryKey() const override;
};
