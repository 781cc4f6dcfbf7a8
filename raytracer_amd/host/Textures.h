// Textures.h -- names of the texture / normal-map functions a Material can point at.
//
// In the reference these are host functions evaluated per hit (Textures.h:22-188).  Here they are *tags*: the
// flattening step (Scene.h) maps each address to an rt_texture_id and the HIP shade kernel evaluates the
// function on the GPU (csrc/rt_textures.h).  Calling one on the host forwards a single uv to the GPU through
// rt_eval_texture(), so `material.texture(uv)` keeps working in host code, e.g. in tests.
#pragma once
#include "rt_math.h"
#include "../../include/rt_hip.h"

glm::vec3 perlinNormal(glm::vec2 uv);
glm::vec3 perlinWater(glm::vec2 uv);
glm::vec3 opal(glm::vec2 uv);
glm::vec3 perlinTerrain(glm::vec2 uv);
glm::vec3 perlinIceTerrain(glm::vec2 uv);
glm::vec3 snowTerrain(glm::vec2 uv);
glm::vec3 qwilfishTexture(glm::vec2 uv);
glm::vec3 colorAt(glm::vec2 uv);
glm::vec3 normalAt(glm::vec2 uv);
float ambientOcclusionAt(glm::vec2 uv);
float roughnessAt(glm::vec2 uv);

// address -> id; returns -1 for a function the GPU does not implement (flattening then fails)
int rt_texture_id_of(glm::vec3 (*f)(glm::vec2));
int rt_texture_id_of(float (*f)(glm::vec2));
// context used by the host-callable forms above (set by render()/upload)
void rt_set_host_texture_context(rt_context *ctx);
