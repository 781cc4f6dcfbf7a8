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
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "Object.h"

// %f of the .obj scanner: what strtof returns, computed without it for the common case.  A decimal with at most 19
// significant digits is m * 10^e with m exact in a u64; for m <= 2^53 and |e| <= 22 one double multiplication or
// division is correctly rounded (both operands exact), and rounding that double to float equals rounding the decimal
// itself unless the double sits exactly on a float rounding midpoint (low 29 mantissa bits == 2^28) -- then, and for
// everything unusual ('+', hex, inf / nan, more digits, big exponents, float denormals / overflow), the caller falls
// back to strtof.  Returns false for "not handled here"; on success *end is where strtof would stop.
inline bool objFastFloat(const char *p, const char **end, float *out) {
    static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                      1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *s = p;
    const bool neg = *s == '-';
    if (neg) ++s;
    uint64_t m = 0;
    int nd = 0, e10 = 0;
    bool any = false;
    for (; *s >= '0' && *s <= '9'; ++s, any = true) {
        if (nd >= 19) return false;
        m = m * 10 + (uint64_t)(*s - '0');
        if (m) nd++;
    }
    if (*s == '.') {
        ++s;
        for (; *s >= '0' && *s <= '9'; ++s, any = true) {
            if (nd >= 19) return false;
            m = m * 10 + (uint64_t)(*s - '0');
            if (m) nd++;
            e10--;
        }
    }
    if (!any) return false;
    if (*s == 'x' || *s == 'X') return false;  // "0x...": strtof reads hexadecimal
    if (*s == 'e' || *s == 'E') {
        const char *t = s + 1;
        const bool eneg = *t == '-';
        if (*t == '-' || *t == '+') ++t;
        if (*t >= '0' && *t <= '9') {  // (no digits: the 'e' is not part of the number)
            int ev = 0;
            for (; *t >= '0' && *t <= '9'; ++t)
                if (ev < 10000) ev = ev * 10 + (*t - '0');
            e10 += eneg ? -ev : ev;
            s = t;
        }
    }
    double d;
    if (m == 0) {
        d = 0.0;
    } else {
        if (m > (1ull << 53) || e10 < -22 || e10 > 22) return false;
        d = e10 < 0 ? (double)m / kPow10[-e10] : (double)m * kPow10[e10];
        if (!(d >= 1.17549435e-38 && d <= 3.40282347e+38)) return false;  // float denormal or overflow: strtof
        uint64_t bits;
        memcpy(&bits, &d, 8);
        if ((bits & 0x1FFFFFFFull) == 0x10000000ull) return false;      // on a float midpoint: strtof decides
    }
    *out = (float)(neg ? -d : d);
    *end = s;
    return true;
}

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
        // The whole file is read at once and scanned in place.  The reference parses every line with sscanf
        // (MeshLoader.h:323-371); sscanf on a string first measures it and sets up a stream, which made the parse 3x
        // slower than the conversions themselves.  The scanner below performs the same conversions (strtof / strtol
        // are what %f / %d call) with sscanf's matching rules: a literal must match exactly, white space in the format
        // skips any white space, %f skips leading white space, the first failure ends the line and leaves the
        // remaining targets as they were.
        // Deliberate difference on MALFORMED numbers (none occur in the reference's assets, so parity is unpinned there,
        // not broken): glibc's scanf consumes a dangling exponent ("1e", "1.5e+" followed by a non-digit) and then FAILS
        // the conversion, leaving the target stale and ending the line; strtof -- and this scanner -- convert the mantissa
        // (1, 1.5) and stop in front of the 'e'.  Integers beyond strtol's range saturate here where %d wraps.
        // tests/test_host_objscan.py: test_dangling_exponent_follows_strtof states the choice.
        std::string buf;
        {
            FILE *fp = fopen(filename.c_str(), "rb");
            if (!fp) throw std::runtime_error("MeshLoader: could not open " + filename);
            char chunk[1 << 16];
            size_t got;
            while ((got = fread(chunk, 1, sizeof chunk, fp)) > 0) buf.append(chunk, got);
            fclose(fp);
        }
        minBounds = glm::vec3(INFINITY);
        maxBounds = glm::vec3(-INFINITY);
        // scan targets live across lines on purpose (stale-index quirk)
        float x = 0, y = 0, z = 0, nx = 0, ny = 0, nz = 0, tx = 0, ty = 0, tz = 0;
        int smoothShading = 0;
        bool sawTexture = false;
        struct Scan {
            const char *p;
            bool ok = true;
            static bool space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
            void ws() { while (space(*p)) ++p; }
            Scan &lit(char c) {   // an ordinary character of the format
                if (ok) { if (*p == c) ++p; else ok = false; }
                return *this;
            }
            Scan &sp() {          // white space in the format
                if (ok) ws();
                return *this;
            }
            Scan &f(float &out) {  // %f
                if (ok) {
                    ws();
                    const char *e;
                    float v;
                    if (!objFastFloat(p, &e, &v)) {
                        char *e2;
                        v = strtof(p, &e2);
                        e = e2;
                    }
                    if (e == p) ok = false; else { out = v; p = e; }
                }
                return *this;
            }
            Scan &d(int &out) {    // %d
                if (ok) {
                    ws();
                    char *e;
                    const long v = strtol(p, &e, 10);
                    if (e == p) ok = false; else { out = (int)v; p = e; }
                }
                return *this;
            }
        };
        char *cur = buf.empty() ? nullptr : &buf[0], *const end = cur + buf.size();
        while (cur && cur < end) {  // std::getline's lines: up to '\n' (a last line without one counts)
            char *nl = (char *)memchr(cur, '\n', (size_t)(end - cur));
            char *const next = nl ? nl + 1 : end;
            if (nl) *nl = '\0';  // (end itself is the string's terminator)
            const size_t len = (size_t)((nl ? nl : end) - cur);
            const char c0 = len > 0 ? cur[0] : '\0';
            const char c1 = len > 1 ? cur[1] : '\0';
            Scan sc{cur};
            if (c1 == 'n') {
                sc.lit('v').lit('n').sp().f(x).sp().f(y).sp().f(z);  // "vn %f %f %f"
                normals.emplace_back(x + translation.x, y + translation.y, z + translation.z);
            } else if (c1 == 't') {
                sc.lit('v').lit('t').sp().f(x).sp().f(y);            // "vt %f %f"
                textureCoords.emplace_back(x + translation.x, y + translation.y);
                sawTexture = true;
            } else if (c0 == 'v') {
                sc.lit('v').sp().f(x).sp().f(y).sp().f(z);           // "v %f %f %f"
                glm::vec3 p(x + translation.x, y + translation.y, z + translation.z);
                vertices.push_back(p);
                for (int k = 0; k < 3; k++) {
                    minBounds[k] = std::fmin(minBounds[k], p[k]);
                    maxBounds[k] = std::fmax(maxBounds[k], p[k]);
                }
            } else if (c0 == 's') {
                sc.lit('s').sp().d(smoothShading);                   // "s %d"
            } else if (c0 == 'f') {
                if (smoothShading == 0) {
                    sc.lit('f').sp().f(x).sp().f(y).sp().f(z);       // "f %f %f %f"
                    addTriangle(at(vertices, x, "vertex", filename), at(vertices, y, "vertex", filename),
                                at(vertices, z, "vertex", filename));
                } else if (sawTexture) {                             // "f %f/%f/%f %f/%f/%f %f/%f/%f"
                    sc.lit('f').sp().f(x).lit('/').f(tx).lit('/').f(nx).sp().f(y).lit('/').f(ty).lit('/').f(ny).sp()
                        .f(z).lit('/').f(tz).lit('/').f(nz);
                    addTriangle(at(vertices, x, "vertex", filename), at(vertices, y, "vertex", filename),
                                at(vertices, z, "vertex", filename), at(normals, nx, "normal", filename),
                                at(normals, ny, "normal", filename), at(normals, nz, "normal", filename),
                                at(textureCoords, tx, "uv", filename), at(textureCoords, ty, "uv", filename),
                                at(textureCoords, tz, "uv", filename));
                } else {                                             // "f %f//%f %f//%f %f//%f"
                    sc.lit('f').sp().f(x).lit('/').lit('/').f(nx).sp().f(y).lit('/').lit('/').f(ny).sp()
                        .f(z).lit('/').lit('/').f(nz);
                    addTriangle(at(vertices, x, "vertex", filename), at(vertices, y, "vertex", filename),
                                at(vertices, z, "vertex", filename), at(normals, nx, "normal", filename),
                                at(normals, ny, "normal", filename), at(normals, nz, "normal", filename));
                }
            }
            cur = next;
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
