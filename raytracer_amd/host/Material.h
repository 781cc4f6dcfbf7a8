// Material.h -- struct Material of the scene-definition surface; field names and defaults follow the
// reference's Material.h:14-42 so existing scene code compiles unchanged.
#pragma once
#include "Textures.h"

struct Material {
    glm::vec3 ambient = glm::vec3(0.0f);
    glm::vec3 diffuse = glm::vec3(1.0f);
    glm::vec3 specular = glm::vec3(0.0f);
    float reflection = 0.0f;
    float refraction = 0.0f;
    float sigma = 1.0f;  // refractive index
    float shininess = 0.0f;
    glm::vec3 (*texture)(glm::vec2 uv) = nullptr;
    bool hasNormalMap = false;
    glm::vec3 (*normalMap)(glm::vec2 uv) = nullptr;
    bool isAnisotropic = false;
    float alpha_x = 0.5f;
    float alpha_y = 0.2f;
    bool hasImgTexture = false;
    float (*roughness)(glm::vec2 uv) = nullptr;
    float (*occlusion)(glm::vec2 uv) = nullptr;
};
