/*
 * trg_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of ToyRaygun's Metal-backend path-tracing hot path
 * (SURVEY.md section 8a, rows a1..a15).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (toyraygun_amd/csrc) never links, includes or calls it.
 *
 * PARITY STATUS: geometry pinned, arithmetic "parity unpinned".
 *   The reference has no tests, golden vectors or CPU render path,
 *   its intersector is closed vendor code (MPS / DXR) and four bx matrix
 *   functions exist only as prebuilt Mach-O / COFF objects.  The reference's own
 *   shader sources are MSL and need <metal_stdlib>/<simd/simd.h>, which this image
 *   lacks; building them would need stand-in headers, so there is no oracle/_ref.
 *   Every function below cites the reference file:line it restates.  The only
 *   reference OUTPUTS are its two README screenshots: tests/test_screenshot_pin.py
 *   pins this restatement's camera, scene matrices, flattening, raygen and image
 *   orientation against both to 0.5 pixel on ~110 step edges (and fails on a 0.5 %
 *   field-of-view error); the intersector's arithmetic, sin/cos rounding and the
 *   shading values have no reference vector and stay unpinned.
 *
 * Floating-point contract of this restatement: IEEE binary32, round-to-nearest,
 * NO contraction (built with -ffp-contract=off), expressions evaluated exactly
 * in the order the reference source text writes them.
 */
#ifndef TRG_ORACLE_H
#define TRG_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* runtime/shaders/common.h:14-34 + src/engine/Uniforms.h:19-41 (176-byte layout;
 * inv_view_proj is the memory image of the simd float4x4 that
 * MetalUniforms.h:49-59 fills with a transpose: m[col*4+row] = bx[row*4+col]). */
typedef struct orc_uniforms {
    uint32_t width, height, frameIndex, _pad;
    float cam_pos[4];
    float inv_view_proj[16];
    float light_pos[4], light_forward[4], light_right[4], light_up[4], light_color[4];
} orc_uniforms;

/* runtime/shaders/metal/Raytracing.metal:21-38 */
typedef struct orc_ray {
    float origin[3];
    uint32_t mask;
    float direction[3];
    float maxDistance;
    float color[4]; /* float3 padded to 16 B */
} orc_ray; /* 48 B */

typedef struct orc_isect {
    float distance;
    int32_t primitiveIndex;
    float coordinates[2]; /* weights of vertex 0 and vertex 1 (MPS convention) */
} orc_isect; /* 16 B */

/* src/engine/Scene.h:25-29: five flat vectors */
typedef struct orc_scene {
    float *positions;  /* nverts*3 */
    float *normals;    /* nverts*3 */
    float *colors;     /* nverts*3 */
    uint32_t *indices; /* nverts   */
    uint32_t *material_ids; /* ntris */
    uint32_t nverts, ntris, cap_verts, cap_tris;
    void *accel; /* lazily built oracle BVH (large scenes only) */
    /* albedo textures -- NOT in the reference ("OBJ and Texture support" is on its to-do list, README.md:18-22): the project's
     * definition, restated here for the parity tests (orc_scene_set_textures) */
    float *uvs;           /* ntris*3 (u,v) pairs, or NULL */
    uint32_t *tex_ids;    /* ntris: 0 = none, k = texture k-1 */
    uint32_t *tex_table;  /* per texture: first texel, width, height, 0 */
    uint32_t *texels;     /* RGBA8, all textures back to back */
    uint32_t ntextures;
    /* an acceleration structure handed in from outside (orc_scene_set_bvh: bench.py's tuned CPU-baseline leg walks THE PRODUCT'S BVH2, as
     * BASELINE.md section 3 plans -- "the same BVH"): nodes of 16 floats (ax0 ax1 ay0 ay1 | bx0 bx1 by0 by1 | az0 az1 bz0 bz1 | child0
     * child1 - -; a child < 0 is the leaf code ~((first << 3) | (count - 1)), count field 7 = a quad leaf of two triangles), records of 12 floats (v0 | original index, e1 | mask, e2 | 0) */
    float *xnodes, *xtris;
    uint32_t xn_nodes, xn_recs;
} orc_scene;
/* copies the arrays (what toyraygun_amd.capi.debug_build_bvh returns); n_nodes = 0 removes them.  While set, orc_intersect_* and orc_render
 * walk this tree instead of brute force / the oracle's own median-split tree: same hits (the triangle arithmetic is tri_test's on the
 * same fp32 v0 / e1 / e2, ties to the lower original index). */
void orc_scene_set_bvh(orc_scene *s, const float *nodes16, uint32_t n_nodes, const float *tris12, uint32_t n_recs);

typedef struct orc_stats {
    uint64_t primary_rays;   /* raygen rays (one per pixel-sample)              */
    uint64_t bounce_rays;    /* nearest-hit rays submitted with maxDistance>=0 at bounce>=1 */
    uint64_t shadow_rays;    /* any-hit rays submitted with maxDistance>=0      */
    uint64_t shaded_hits;    /* primaryHit invocations that took the MATERIAL_DEFAULT branch */
    uint64_t tri_tests;      /* ray/triangle tests executed by the oracle intersector */
    uint64_t node_visits;    /* BVH2-equivalent boxes tested by the oracle intersector */
} orc_stats;

enum { ORC_TRIG_LIBM = 0, ORC_TRIG_PORTABLE = 1 };
/* ORC_TRIG_LIBM: cos()/sin() of common.h:84-85 are libm cosf/sinf (faithful default).
 * ORC_TRIG_PORTABLE: a fixed polynomial written with plain IEEE ops only, which the HIP
 * kernel restates operation by operation, so that strict builds can be compared bit for bit. */
void orc_set_trig_mode(int mode);
/* NEGATIVE CONTROLS for tests/test_screenshot_pin.py (radiometric pin): estimators the reference's Metal backend does NOT implement.
 * 0 (default) = Raytracing.metal / common.h as written.  ORC_VAR_BOUNCE_SEES_LIGHT: continuation rays keep RAY_MASK_PRIMARY, so they
 * can hit the emissive quad (what the D3D12 shader allows, Raytracing.hlsl:227-228; Raytracing.metal:198 gives them mask 1).
 * ORC_VAR_LIGHT_NORMALISED: the light sample is weighted by area / pi (a radiometrically normalised area light) instead of
 * common.h:119-159's bare 1 / max(dist, 1e-3)^2. */
enum { ORC_VAR_BOUNCE_SEES_LIGHT = 1, ORC_VAR_LIGHT_NORMALISED = 2 };
void orc_set_variant(int flags);
int  orc_get_trig_mode(void);

/* ---- bx matrix functions (bodies absent from the reference tree; restated from the
 *      published bkaradzic/bx src/math.cpp algorithm, version unpinned: SURVEY 8c) ---- */
void orc_mtx_srt(float *m, float sx, float sy, float sz, float ax, float ay, float az,
                 float tx, float ty, float tz);
void orc_mtx_look_at_rh(float *m, const float eye[3], const float at[3], const float up[3]);
void orc_mtx_proj_rh(float *m, float fovy_deg, float aspect, float near_, float far_);
void orc_vec4_mul_mtx(float *r, const float *v, const float *m);
void orc_mtx_mul(float *r, const float *a, const float *b);
void orc_mtx_inverse(float *r, const float *a);

/* ---- scene (a1) ---- */
orc_scene *orc_scene_new(void);
void orc_scene_free(orc_scene *s);
void orc_scene_add_geometry(orc_scene *s, const float *verts3, const uint32_t *tri_idx, int tri_count,
                            const float *mtx, const float color[3], uint32_t material_id);
void orc_scene_add_raw(orc_scene *s, const float *pos9, const float *nrm9, const float *col9, const uint32_t *material_ids,
                       int tri_count); /* Scene.h:25-29 public vectors, appended as given */
/* project definition (see orc_scene above): uv2 = ntris*3 pairs, tex_ids = ntris, images = ntextures pointers to RGBA8 rows */
void orc_scene_set_textures(orc_scene *s, const float *uv2, const uint32_t *tex_ids, const uint8_t *const *images_rgba8,
                            const uint32_t *widths, const uint32_t *heights, uint32_t ntextures);
void orc_scene_add_cube(orc_scene *s, const float color[3], const float *mtx);
void orc_scene_add_plane(orc_scene *s, const float color[3], const float *mtx);
void orc_scene_add_area_light(orc_scene *s, const float color[3], const float *mtx);
void orc_scene_cornell_box(orc_scene *s);
/* Cornell box + the reference cube mesh replicated on an n^3 lattice */
void orc_scene_cornell_lattice(orc_scene *s, int n, uint32_t seed);

/* ---- camera + uniforms (a2, a3) ---- */
void orc_make_uniforms(orc_uniforms *u, uint32_t w, uint32_t h, uint32_t frame_index,
                       const float eye[3], const float at[3]);

/* ---- per-pixel Halton index offsets (a4; seeded replacement for libc rand()) ---- */
uint32_t orc_pcg_hash32(uint32_t v);
void orc_pixel_offsets(uint32_t seed, uint32_t w, uint32_t h, uint32_t *out);

/* ---- sampling (a5, a10, a11) ---- */
#define ORC_HALTON_DIMS 64
uint32_t orc_halton_prime(uint32_t d);
float orc_halton(uint32_t i, uint32_t d);
void orc_sample_cosine_hemisphere(const float u[2], float out[3]);
void orc_align_hemisphere(const float sample[3], const float normal[3], float out[3]);
void orc_sample_area_light(const orc_uniforms *u, const float r[2], const float pos[3],
                           const float nrm[3], float dir[3], float color[3], float *dist);
void orc_sincos_portable(float phi, float *s, float *c);

/* ---- the four kernels, one invocation per pixel (a6, a9, a13, a15) ---- */
void orc_raygen(const orc_uniforms *u, const uint32_t *offsets, orc_ray *rays, float *dst_rgba,
                uint32_t x, uint32_t y);
void orc_primary_hit(const orc_uniforms *u, orc_ray *rays, orc_ray *shadow_rays, const orc_isect *isects,
                     const orc_scene *s, uint32_t bounce, const uint32_t *offsets, float *dst_rgba,
                     uint32_t x, uint32_t y, orc_stats *st);
void orc_shadow_hit(const orc_uniforms *u, const orc_ray *shadow_rays, const float *shadow_dist,
                    const float *src_rgba, float *dst_rgba, uint32_t x, uint32_t y);
void orc_accumulate(const orc_uniforms *u, const float *render_rgba, const float *prev_rgba,
                    float *accum_rgba, uint32_t x, uint32_t y);

/* ---- intersector (a7, a12): project definition, see DESIGN.md "intersection contract" ---- */
void orc_intersect_nearest(orc_scene *s, const orc_ray *rays, size_t n, orc_isect *out, orc_stats *st);
void orc_intersect_any(orc_scene *s, const orc_ray *rays, size_t n, float *out_dist, orc_stats *st);
/* brute force regardless of scene size (checks the oracle's own BVH) */
void orc_intersect_nearest_brute(const orc_scene *s, const orc_ray *rays, size_t n, orc_isect *out);
void orc_intersect_any_brute(const orc_scene *s, const orc_ray *rays, size_t n, float *out_dist);
/* double-precision geometric ground truth for one ray: returns primitive index or -1 */
int orc_nearest_f64(const orc_scene *s, const orc_ray *ray, double *t_out, double *margin_out);

/* ---- frame loop (a14): renders frames [frame_begin, frame_begin+spp) for rows [row0,row0+rows)
 *      into the full-frame float4 accumulation buffer accum (w*h*4, row 0 = scene bottom).
 *      If frame_begin > 0, accum must hold the running average of frames [0, frame_begin).
 *      nthreads<=0 -> all cores.  Returns 0. */
int orc_render(orc_scene *s, const orc_uniforms *u_base, const uint32_t *offsets,
               uint32_t frame_begin, uint32_t spp, uint32_t bounces,
               uint32_t row0, uint32_t rows, float *accum, orc_stats *st, int nthreads);

/* ---- post-process (N1): ACES + sRGB, common.h:36-43,163-171; PostProcessing.metal:44-57 ---- */
void orc_postprocess(const float *accum_rgba, uint32_t w, uint32_t h, uint8_t *rgba8_out, int flip_y);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
