// Objects.h -- analytic primitives of the scene-definition surface: Sphere, Plane and Cone with the constructors
// of the reference (Objects.h:25,31,127,198).
#pragma once
#include "Object.h"

class Sphere : public Object {
public:
    explicit Sphere(glm::vec3 c) { color = c; }
    explicit Sphere(Material m) { material = m; }
    void flatten(FlatScene &out, int object_id) const override;
};

class Plane : public Object {
    glm::vec3 normal = glm::vec3(0.0f, 1.0f, 0.0f);
    glm::vec3 point = glm::vec3(0.0f);

public:
    Plane(glm::vec3 point_, glm::vec3 normal_, bool hasMaterial, Material m = Material())
        : normal(normal_), point(point_) {
        if (hasMaterial) material = m;
    }
    glm::vec3 getNormal() const { return normal; }
    glm::vec3 getPoint() const { return point; }
    void flatten(FlatScene &out, int object_id) const override;
    uint64_t geometryKey() const override;
};

// Unit cone (apex at the origin, opening towards +y, closed by the disc y = 1) under the object's transformation
// (reference Objects.h:188-278; the base Plane the reference allocates is part of the GPU-side intersection).
class Cone : public Object {
public:
    explicit Cone(Material m) { material = m; }
    void flatten(FlatScene &out, int object_id) const override;
};
