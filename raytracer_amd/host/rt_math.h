// rt_math.h -- the small slice of vector/matrix maths the scene-definition surface needs on the host.
//
// The reference describes scenes with glm types (glm::vec3, glm::mat4, glm::translate/scale/rotate,
// Object::setTransformation -> glm::inverse / glm::transpose; reference Object.h:53-57, main.cpp:362-400,
// 523-529, 574-578).  This header restates only those operations, with the same order of floating-point
// operations as glm 0.9.9.7 (cited per function) so that the matrices handed to the GPU are bit-identical to
// the reference's.  It is NOT a glm clone: nothing here runs on the render path.
//
// Scene code written against glm keeps compiling: unless real glm headers were included first, `glm` is an
// alias of this namespace.
#pragma once
#include <cmath>
#include <cstddef>

namespace rtm {

struct vec2 {
    float x = 0, y = 0;
    vec2() = default;
    explicit vec2(float s) : x(s), y(s) {}
    vec2(float x_, float y_) : x(x_), y(y_) {}
    vec2(double x_, double y_) : x((float)x_), y((float)y_) {}
};

struct vec3 {
    union { float x; float r; };
    union { float y; float g; };
    union { float z; float b; };
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    explicit vec3(double s) : x((float)s), y((float)s), z((float)s) {}
    explicit vec3(int s) : x((float)s), y((float)s), z((float)s) {}
    template <class A, class B, class C>
    vec3(A x_, B y_, C z_) : x((float)x_), y((float)y_), z((float)z_) {}
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    vec4() = default;
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    vec4(const vec3 &v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};

inline vec3 operator+(const vec3 &a, const vec3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(const vec3 &a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(const vec3 &a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, const vec3 &a) { return {a.x * s, a.y * s, a.z * s}; }
inline bool operator==(const vec3 &a, const vec3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(const vec3 &a, const vec3 &b) { return !(a == b); }
inline bool operator==(const vec2 &a, const vec2 &b) { return a.x == b.x && a.y == b.y; }
inline bool operator!=(const vec2 &a, const vec2 &b) { return !(a == b); }
inline vec4 operator*(const vec4 &a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator+(const vec4 &a, const vec4 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator-(const vec4 &a, const vec4 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline vec4 operator*(const vec4 &a, const vec4 &b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }

// glm/detail/func_geometric.inl:48-55 (dot, left-to-right sum), :70-78 (cross), :82-90 (normalize = v * 1/sqrt)
inline float dot(const vec3 &a, const vec3 &b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
inline vec3 cross(const vec3 &a, const vec3 &b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline vec3 normalize(const vec3 &v) { return v * (1.0f / std::sqrt(dot(v, v))); }

// column-major 4x4, m[c] is column c (glm convention)
struct mat4 {
    vec4 c[4];
    mat4() = default;
    explicit mat4(float d) {
        c[0] = {d, 0, 0, 0};
        c[1] = {0, d, 0, 0};
        c[2] = {0, 0, d, 0};
        c[3] = {0, 0, 0, d};
    }
    vec4 &operator[](int i) { return c[i]; }
    const vec4 &operator[](int i) const { return c[i]; }
};

// glm/detail/type_mat4x4.inl:630-648
inline mat4 operator*(const mat4 &a, const mat4 &b) {
    mat4 r;
    for (int k = 0; k < 4; k++) r[k] = a[0] * b[k][0] + a[1] * b[k][1] + a[2] * b[k][2] + a[3] * b[k][3];
    return r;
}
// glm/detail/type_mat4x4.inl:561-572: (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
inline vec4 operator*(const mat4 &m, const vec4 &v) {
    vec4 add0 = m[0] * v.x + m[1] * v.y;
    vec4 add1 = m[2] * v.z + m[3] * v.w;
    return add0 + add1;
}

// glm/ext/matrix_transform.inl:10-15 (translate), :78-86 (scale), :18-46 (rotate)
inline mat4 translate(const mat4 &m, const vec3 &v) {
    mat4 r = m;
    r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
    return r;
}
inline mat4 scale(const mat4 &m, const vec3 &v) {
    mat4 r;
    r[0] = m[0] * v.x;
    r[1] = m[1] * v.y;
    r[2] = m[2] * v.z;
    r[3] = m[3];
    return r;
}
inline mat4 translate(const vec3 &v) { return translate(mat4(1.0f), v); }  // glm/gtx/transform.inl
inline mat4 scale(const vec3 &v) { return scale(mat4(1.0f), v); }
inline mat4 rotate(const mat4 &m, float angle, const vec3 &axis_in) {
    const float c = std::cos(angle), s = std::sin(angle);
    vec3 axis = normalize(axis_in);
    vec3 temp = (1.0f - c) * axis;
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = temp.x * axis.y + s * axis.z;
    R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y;
    R[2][1] = temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    mat4 r;
    for (int k = 0; k < 3; k++) r[k] = m[0] * R[k][0] + m[1] * R[k][1] + m[2] * R[k][2];
    r[3] = m[3];
    return r;
}

inline mat4 transpose(const mat4 &m) {
    mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[i][j] = m[j][i];
    return r;
}

// glm/detail/func_matrix.inl:294-352 -- cofactor inverse, same sub-expression order
inline mat4 inverse(const mat4 &m) {
    float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    vec4 f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11);
    vec4 f3(c12, c12, c14, c15), f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
    vec4 v0(m[1][0], m[0][0], m[0][0], m[0][0]), v1(m[1][1], m[0][1], m[0][1], m[0][1]);
    vec4 v2(m[1][2], m[0][2], m[0][2], m[0][2]), v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    vec4 i0 = v1 * f0 - v2 * f1 + v3 * f2;
    vec4 i1 = v0 * f0 - v2 * f3 + v3 * f4;
    vec4 i2 = v0 * f1 - v1 * f3 + v3 * f5;
    vec4 i3 = v0 * f2 - v1 * f4 + v2 * f5;
    vec4 sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
    mat4 inv;
    inv[0] = i0 * sa;
    inv[1] = i1 * sb;
    inv[2] = i2 * sa;
    inv[3] = i3 * sb;
    vec4 row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    vec4 d = m[0] * row0;
    float det = (d.x + d.y) + (d.z + d.w);
    float ood = 1.0f / det;
    for (int k = 0; k < 4; k++) inv[k] = inv[k] * ood;
    return inv;
}

}  // namespace rtm

#ifndef GLM_VERSION
namespace glm = rtm;
#endif
