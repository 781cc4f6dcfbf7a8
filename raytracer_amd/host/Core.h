// Core.h -- Ray and Hit carriers of the scene-definition surface (host side).
// Same names and fields as the reference's Core.h:12-57 so scene / test code reads the same; the GPU path
// uses its own packed records (csrc/rt_device.h) and fills `Hit` only at the rt_trace_batch boundary.
#pragma once
#include "rt_math.h"

class Object;

class Ray {
public:
    glm::vec3 origin;
    glm::vec3 direction;
    Ray(glm::vec3 o, glm::vec3 d) : origin(o), direction(d) {}
};

struct Hit {
    bool hit = false;
    glm::vec3 normal;         // geometric (or vertex-interpolated) normal
    glm::vec3 intersection;   // world-space point
    float distance = 0;
    Object *object = nullptr;
    glm::vec2 uv;
    glm::vec3 normalShading;  // after normal mapping
    glm::vec3 tangent;
    glm::vec3 bitangent;
    int triangle = -1;        // extension: triangle index inside a mesh, -1 otherwise
};
