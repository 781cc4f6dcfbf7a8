/* rt_oracle.h -- C interface of the CPU ORACLE (librt_oracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported baseline -- never as the thing measured or shipped.  The product
 * (raytracer_amd/, libraytracer_hip.so, libraytracer_host.so) neither links nor imports anything from oracle/.
 *
 * It is a CPU restatement of the reference's trace loop (see rt_oracle.cpp for per-function citations) over the
 * same flattened scene arrays the C ABI takes (include/rt_hip.h), so one set of inputs feeds both sides.
 * Pinned against the compiled reference itself (oracle/_ref, fixtures in tests/golden/): see tests/test_oracle_pin.py.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include "../include/rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ro_scene ro_scene;

typedef struct ro_scene_desc {
    int32_t n_materials;
    const rt_material *materials;
    int32_t n_objects;
    const int32_t *object_material;
    int64_t n_triangles;
    const float *v0, *v1, *v2, *nA, *nB, *nC, *tA, *tB, *tC;
    const uint32_t *tri_flags;
    const int32_t *tri_object;
    int32_t n_spheres;
    const float *sph_M, *sph_Minv, *sph_N;
    const int32_t *sph_object;
    int32_t n_planes;
    const float *pl_point, *pl_normal;
    const int32_t *pl_object;
    int32_t n_lights;
    const float *light_pos, *light_color;
    float ambient[3];
    /* BMP slots 0..3 (color, normal, roughness, occlusion), top-down BGR; width 0 = absent */
    int32_t bmp_w[4], bmp_h[4], bmp_c[4];
    const uint8_t *bmp[4];
    /* cones (Objects.h:188-278): same three matrices as spheres */
    int32_t n_cones;
    const float *cone_M, *cone_Minv, *cone_N;
    const int32_t *cone_object;
} ro_scene_desc;

ro_scene *ro_create(const ro_scene_desc *d);
void ro_destroy(ro_scene *s);

/* Same semantics as rt_render (rgb8 as bytes; rgb8_i32 additionally keeps the reference's raw ints, may be NULL) */
int ro_render(const ro_scene *s, const rt_camera *cam, const rt_render_params *p, uint8_t *out_rgb8,
              int32_t *out_rgb8_i32, float *out_rgbf, uint32_t *out_primid, rt_stats *stats, int threads);
int ro_trace_batch(const ro_scene *s, int64_t n, const float *rays, int32_t depth, float *out_color,
                   rt_hit *out_hits, uint8_t *out_shadow, rt_stats *stats, int threads);
int ro_eval_texture(const ro_scene *s, int32_t texture_id, int64_t n, const float *uv, float *out);
int ro_eval_perlin(int64_t n, const float *p, float *out);
/* rank[t] = position of triangle t in the order the reference's per-mesh BVH hands candidates out
 * (MeshLoader.h:265-284: left subtree before right, file order inside a leaf), meshes in object order */
int ro_reference_order(const ro_scene *s, uint32_t *rank);
/* test aid: from now on every closest-hit ray (primary rays and bounces of ro_render / ro_trace_batch) is written to `path`,
 * one line of hex float bits per ray: origin, direction, then hit, object, triangle (scene numbering, -1 = none), distance.
 * NULL or "" stops.  Process-wide; not for concurrent scenes. */
int ro_set_ray_log(const char *path);
#ifdef __cplusplus
}
#endif
#endif
