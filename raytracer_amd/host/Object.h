// Object.h -- base of every scene object (reference Object.h:10-59): the three transformation matrices,
// colour and material.  Objects are *descriptions*: intersection runs on the GPU (csrc/rt_kernels.hip), so
// instead of a virtual intersect() each object knows how to append itself to a FlatScene.
#pragma once
#include <cstdint>
#include "Core.h"
#include "Material.h"

struct FlatScene;

class Object {
protected:
    glm::mat4 transformationMatrix = glm::mat4(1.0f);
    glm::mat4 inverseTransformationMatrix = glm::mat4(1.0f);
    glm::mat4 normalMatrix = glm::mat4(1.0f);

public:
    glm::vec3 color;
    Material material;
    virtual ~Object() = default;

    Material getMaterial() const { return material; }
    void setMaterial(Material m) { material = m; }
    // Object.h:53-57: inverse and transpose(inverse) are derived here, bit-identically (rt_math.h)
    void setTransformation(glm::mat4 matrix) {
        transformationMatrix = matrix;
        inverseTransformationMatrix = glm::inverse(matrix);
        normalMatrix = glm::transpose(inverseTransformationMatrix);
    }
    const glm::mat4 &getTransformation() const { return transformationMatrix; }
    const glm::mat4 &getInverseTransformation() const { return inverseTransformationMatrix; }
    const glm::mat4 &getNormalMatrix() const { return normalMatrix; }

    // appends this object's primitives; object_id is its position in the scene's object list
    virtual void flatten(FlatScene &out, int object_id) const = 0;
    // what of the object's own geometry the uploaded-scene fingerprint must see beyond material and transformation
    virtual uint64_t geometryKey() const { return 0; }
};
