/*
 * trg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See trg_oracle.h.
 * PARITY STATUS: the reference holds no tests and no golden vectors; its only outputs are the two README screenshots,
 * and this restatement is PINNED to them: silhouettes / creases / light quad to 0.5 px on both (tests/test_screenshot_pin.py, geometry),
 * colours to 1/255 on 16 flat patches and over all ~8,000 single-face 8x8 cells of the Metal screenshot (radiometry; the D3D12 picture,
 * another estimator, fails the same tests).  Still UNPINNED, because nothing the reference ships can see them: the closed MPS / DXR
 * intersector's arithmetic at record level (ties, t ~ 0, edges: the project's own INTERSECTION CONTRACT below) and the rounding of sin / cos.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (oracle/Makefile).
 * All citations are file:line under /root/reference.
 */
#include "trg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* =============================== small vector helpers =============================== */
/* Evaluation order mirrors MSL/bx component-wise semantics: dot = x*x + y*y + z*z (left to right). */
static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(const float a[3], const float b[3], float r[3])
{
    /* bx inline/math.inl:620-628 and MSL cross() */
    float x = a[1] * b[2] - a[2] * b[1];
    float y = a[2] * b[0] - a[0] * b[2];
    float z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
/* normalize(v) = v * (1 / sqrt(dot(v,v)))  -- bx inline/math.inl:666-671; MSL normalize() is
 * v * rsqrt(dot(v,v)), whose rounding is implementation-defined (unpinned). */
static inline void normalize3(const float a[3], float r[3])
{
    float inv = 1.0f / sqrtf(dot3(a, a));
    r[0] = a[0] * inv; r[1] = a[1] * inv; r[2] = a[2] * inv;
}
static inline float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

/* =============================== trig mode =============================== */
static int g_trig_mode = ORC_TRIG_LIBM;
void orc_set_trig_mode(int mode) { g_trig_mode = mode; }
int orc_get_trig_mode(void) { return g_trig_mode; }
static int g_variant = 0;   /* negative controls of the radiometric screenshot pin (trg_oracle.h); 0 = the reference's estimator */
void orc_set_variant(int flags) { g_variant = flags; }

/* Portable sin/cos for phi in [0, 2*pi]: quadrant reduction with a 3-term Cody-Waite split of pi/2,
 * then odd/even minimax polynomials on [-pi/4, pi/4].  Plain IEEE ops only, fixed order; the HIP
 * kernel restates it operation for operation (toyraygun_amd/csrc/trg_device.h: trg_sincos). */
void orc_sincos_portable(float phi, float *s_out, float *c_out)
{
    const float two_over_pi = 0x1.45f306p-1f;
    const float pio2_hi = 0x1.92p+0f;       /* 12 significant bits: k*pio2_hi is exact for k <= 4 */
    const float pio2_mid = 0x1.fb5444p-12f;
    const float pio2_lo = 0x1.68c2p-39f;
    float kf = rintf(phi * two_over_pi);
    int k = (int)kf;
    float r = phi - kf * pio2_hi;
    r = r - kf * pio2_mid;
    r = r - kf * pio2_lo;
    float z = r * r;
    /* Cephes sinf/cosf minimax polynomials on [-pi/4, pi/4] */
    float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    float sn = (ps * z) * r + r;
    float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    float cs = ((pc * z) * z - 0.5f * z) + 1.0f;
    float s, c;
    switch (k & 3) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
    }
    *s_out = s; *c_out = c;
}

/* =============================== bx matrix restatements =============================== */
/* bx convention: row-major float[16], row-vector * matrix (inline/math.inl:1228-1242). */
void orc_vec4_mul_mtx(float *r, const float *v, const float *m)
{
    /* lib/bx/include/bx/inline/math.inl:1228-1234 */
    float x = v[0] * m[0] + v[1] * m[4] + v[2] * m[8] + v[3] * m[12];
    float y = v[0] * m[1] + v[1] * m[5] + v[2] * m[9] + v[3] * m[13];
    float z = v[0] * m[2] + v[1] * m[6] + v[2] * m[10] + v[3] * m[14];
    float w = v[0] * m[3] + v[1] * m[7] + v[2] * m[11] + v[3] * m[15];
    r[0] = x; r[1] = y; r[2] = z; r[3] = w;
}
void orc_mtx_mul(float *r, const float *a, const float *b)
{
    /* inline/math.inl:1236-1242 */
    float t[16];
    orc_vec4_mul_mtx(&t[0], &a[0], b);
    orc_vec4_mul_mtx(&t[4], &a[4], b);
    orc_vec4_mul_mtx(&t[8], &a[8], b);
    orc_vec4_mul_mtx(&t[12], &a[12], b);
    memcpy(r, t, sizeof(t));
}
void orc_mtx_srt(float *m, float sx_, float sy_, float sz_, float ax, float ay, float az,
                 float tx, float ty, float tz)
{
    /* declared lib/bx/include/bx/math.h:681-692; body = bkaradzic/bx src/math.cpp mtxSRT
     * (scale, then rotate X*Y*Z, then translate; row-vector convention).  bx's own sin/cos are
     * polynomial approximations; libm is used here (difference <= 1e-7, unpinned). */
    float sx = sinf(ax), cx = cosf(ax);
    float sy = sinf(ay), cy = cosf(ay);
    float sz = sinf(az), cz = cosf(az);
    float sxsz = sx * sz;
    float cycz = cy * cz;
    m[0] = sx_ * (cycz - sxsz * sy);
    m[1] = sx_ * -cx * sz;
    m[2] = sx_ * (cz * sy + cy * sxsz);
    m[3] = 0.0f;
    m[4] = sy_ * (cz * sx * sy + cy * sz);
    m[5] = sy_ * cx * cz;
    m[6] = sy_ * (sy * sz - cycz * sx);
    m[7] = 0.0f;
    m[8] = sz_ * -cx * sy;
    m[9] = sz_ * sx;
    m[10] = sz_ * cx * cy;
    m[11] = 0.0f;
    m[12] = tx; m[13] = ty; m[14] = tz; m[15] = 1.0f;
}
void orc_mtx_look_at_rh(float *m, const float eye[3], const float at[3], const float up_[3])
{
    /* declared math.h:572-578; called src/engine/Renderer.cpp:86 with Handness::Right */
    float d[3] = { eye[0] - at[0], eye[1] - at[1], eye[2] - at[2] };
    float view[3]; normalize3(d, view);
    float uxv[3]; cross3(up_, view, uxv);
    float right[3]; normalize3(uxv, right);
    float up[3]; cross3(view, right, up);
    m[0] = right[0]; m[1] = up[0]; m[2] = view[0]; m[3] = 0.0f;
    m[4] = right[1]; m[5] = up[1]; m[6] = view[1]; m[7] = 0.0f;
    m[8] = right[2]; m[9] = up[2]; m[10] = view[2]; m[11] = 0.0f;
    m[12] = -dot3(right, eye);
    m[13] = -dot3(up, eye);
    m[14] = -dot3(view, eye);
    m[15] = 1.0f;
}
void orc_mtx_proj_rh(float *m, float fovy_deg, float aspect, float near_, float far_)
{
    /* declared math.h:604-612; called Renderer.cpp:87 with homogeneousNdc=true, Handness::Right.
     * toRad: inline/math.inl:17-20 (deg * kPi / 180). */
    const float kPi = 3.1415926535897932384626433832795f;
    float rad = fovy_deg * kPi / 180.0f;
    float height = 1.0f / tanf(rad * 0.5f);
    float width = height * 1.0f / aspect;
    float diff = far_ - near_;
    float aa = (far_ + near_) / diff;
    float bb = (2.0f * far_ * near_) / diff;
    memset(m, 0, sizeof(float) * 16);
    m[0] = width;
    m[5] = height;
    m[8] = 0.0f;
    m[9] = 0.0f;
    m[10] = -aa;
    m[11] = -1.0f;
    m[14] = -bb;
}
void orc_mtx_inverse(float *r, const float *a)
{
    /* declared math.h:716; body = bx src/math.cpp mtxInverse (cofactor expansion) */
    float xx = a[0], xy = a[1], xz = a[2], xw = a[3];
    float yx = a[4], yy = a[5], yz = a[6], yw = a[7];
    float zx = a[8], zy = a[9], zz = a[10], zw = a[11];
    float wx = a[12], wy = a[13], wz = a[14], ww = a[15];
    float det = 0.0f;
    det += xx * (yy * (zz * ww - zw * wz) - yz * (zy * ww - zw * wy) + yw * (zy * wz - zz * wy));
    det -= xy * (yx * (zz * ww - zw * wz) - yz * (zx * ww - zw * wx) + yw * (zx * wz - zz * wx));
    det += xz * (yx * (zy * ww - zw * wy) - yy * (zx * ww - zw * wx) + yw * (zx * wy - zy * wx));
    det -= xw * (yx * (zy * wz - zz * wy) - yy * (zx * wz - zz * wx) + yz * (zx * wy - zy * wx));
    float invDet = 1.0f / det;
    float t[16];
    t[0] = +(yy * (zz * ww - wz * zw) - yz * (zy * ww - wy * zw) + yw * (zy * wz - wy * zz)) * invDet;
    t[1] = -(xy * (zz * ww - wz * zw) - xz * (zy * ww - wy * zw) + xw * (zy * wz - wy * zz)) * invDet;
    t[2] = +(xy * (yz * ww - wz * yw) - xz * (yy * ww - wy * yw) + xw * (yy * wz - wy * yz)) * invDet;
    t[3] = -(xy * (yz * zw - zz * yw) - xz * (yy * zw - zy * yw) + xw * (yy * zz - zy * yz)) * invDet;
    t[4] = -(yx * (zz * ww - wz * zw) - yz * (zx * ww - wx * zw) + yw * (zx * wz - wx * zz)) * invDet;
    t[5] = +(xx * (zz * ww - wz * zw) - xz * (zx * ww - wx * zw) + xw * (zx * wz - wx * zz)) * invDet;
    t[6] = -(xx * (yz * ww - wz * yw) - xz * (yx * ww - wx * yw) + xw * (yx * wz - wx * yz)) * invDet;
    t[7] = +(xx * (yz * zw - zz * yw) - xz * (yx * zw - zx * yw) + xw * (yx * zz - zx * yz)) * invDet;
    t[8] = +(yx * (zy * ww - wy * zw) - yy * (zx * ww - wx * zw) + yw * (zx * wy - wx * zy)) * invDet;
    t[9] = -(xx * (zy * ww - wy * zw) - xy * (zx * ww - wx * zw) + xw * (zx * wy - wx * zy)) * invDet;
    t[10] = +(xx * (yy * ww - wy * yw) - xy * (yx * ww - wx * yw) + xw * (yx * wy - wx * yy)) * invDet;
    t[11] = -(xx * (yy * zw - zy * yw) - xy * (yx * zw - zx * yw) + xw * (yx * zy - zx * yy)) * invDet;
    t[12] = -(yx * (zy * wz - wy * zz) - yy * (zx * wz - wx * zz) + yz * (zx * wy - wx * zy)) * invDet;
    t[13] = +(xx * (zy * wz - wy * zz) - xy * (zx * wz - wx * zz) + xz * (zx * wy - wx * zy)) * invDet;
    t[14] = -(xx * (yy * wz - wy * yz) - xy * (yx * wz - wx * yz) + xz * (yx * wy - wx * yy)) * invDet;
    t[15] = +(xx * (yy * zz - zy * yz) - xy * (yx * zz - zx * yz) + xz * (yx * zy - zx * yy)) * invDet;
    memcpy(r, t, sizeof(t));
}

/* =============================== scene (a1) =============================== */
orc_scene *orc_scene_new(void) { return (orc_scene *)calloc(1, sizeof(orc_scene)); }

static void accel_free(void *a);
void orc_scene_free(orc_scene *s)
{
    if (!s) return;
    free(s->positions); free(s->normals); free(s->colors); free(s->indices); free(s->material_ids);
    free(s->uvs); free(s->tex_ids); free(s->tex_table); free(s->texels);
    accel_free(s->accel);
    free(s->xnodes); free(s->xtris);
    free(s);
}
/* Albedo textures: the project's definition (the reference has none).  Copies everything; ntextures = 0 removes them. */
void orc_scene_set_textures(orc_scene *s, const float *uv2, const uint32_t *tex_ids, const uint8_t *const *images_rgba8,
                            const uint32_t *widths, const uint32_t *heights, uint32_t ntextures)
{
    free(s->uvs); free(s->tex_ids); free(s->tex_table); free(s->texels);
    s->uvs = NULL; s->tex_ids = NULL; s->tex_table = NULL; s->texels = NULL; s->ntextures = 0;
    if (!ntextures) return;
    size_t total = 0;
    for (uint32_t k = 0; k < ntextures; ++k) total += (size_t)widths[k] * heights[k];
    s->uvs = (float *)malloc((size_t)s->ntris * 6 * sizeof(float));
    s->tex_ids = (uint32_t *)malloc((size_t)s->ntris * sizeof(uint32_t));
    s->tex_table = (uint32_t *)malloc((size_t)ntextures * 4 * sizeof(uint32_t));
    s->texels = (uint32_t *)malloc(total * sizeof(uint32_t));
    memcpy(s->uvs, uv2, (size_t)s->ntris * 6 * sizeof(float));
    memcpy(s->tex_ids, tex_ids, (size_t)s->ntris * sizeof(uint32_t));
    size_t first = 0;
    for (uint32_t k = 0; k < ntextures; ++k) {
        s->tex_table[k * 4 + 0] = (uint32_t)first; s->tex_table[k * 4 + 1] = widths[k]; s->tex_table[k * 4 + 2] = heights[k]; s->tex_table[k * 4 + 3] = 0;
        memcpy(&s->texels[first], images_rgba8[k], (size_t)widths[k] * heights[k] * 4);
        first += (size_t)widths[k] * heights[k];
    }
    s->ntextures = ntextures;
}
static void scene_reserve(orc_scene *s, uint32_t more_tris)
{
    uint32_t need_t = s->ntris + more_tris;
    if (need_t > s->cap_tris) {
        uint32_t cap = s->cap_tris ? s->cap_tris : 64;
        while (cap < need_t) cap *= 2;
        s->positions = (float *)realloc(s->positions, (size_t)cap * 9 * sizeof(float));
        s->normals = (float *)realloc(s->normals, (size_t)cap * 9 * sizeof(float));
        s->colors = (float *)realloc(s->colors, (size_t)cap * 9 * sizeof(float));
        s->indices = (uint32_t *)realloc(s->indices, (size_t)cap * 3 * sizeof(uint32_t));
        s->material_ids = (uint32_t *)realloc(s->material_ids, (size_t)cap * sizeof(uint32_t));
        s->cap_tris = cap; s->cap_verts = cap * 3;
    }
}
/* src/engine/Scene.cpp:94-100 applyTransform */
static void apply_transform(const float in[3], const float *mtx, float w, float out[3])
{
    float p[4] = { in[0], in[1], in[2], w }, t[4];
    orc_vec4_mul_mtx(t, p, mtx);
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}
/* src/engine/Scene.cpp:102-129 addGeometry */
void orc_scene_add_geometry(orc_scene *s, const float *verts3, const uint32_t *tri_idx, int tri_count,
                            const float *mtx, const float color[3], uint32_t material_id)
{
    scene_reserve(s, (uint32_t)tri_count);
    if (s->accel) { accel_free(s->accel); s->accel = NULL; }
    for (int i = 0; i < tri_count; ++i) {
        uint32_t idx[3] = { tri_idx[i * 3 + 0], tri_idx[i * 3 + 1], tri_idx[i * 3 + 2] };
        /* bx::calcNormal (inline/math.inl:1264-1271) on the UNtransformed vertices */
        const float *va = &verts3[idx[0] * 3], *vb = &verts3[idx[1] * 3], *vc = &verts3[idx[2] * 3];
        float ba[3] = { vb[0] - va[0], vb[1] - va[1], vb[2] - va[2] };
        float ca[3] = { vc[0] - va[0], vc[1] - va[1], vc[2] - va[2] };
        float bxc[3]; cross3(ba, ca, bxc);
        float normal[3]; normalize3(bxc, normal);
        for (int j = 0; j < 3; ++j) {
            float xv[3], xn[3], xnn[3];
            apply_transform(&verts3[idx[j] * 3], mtx, 1.0f, xv);
            apply_transform(normal, mtx, 0.0f, xn);
            normalize3(xn, xnn);
            uint32_t v = s->nverts;
            memcpy(&s->positions[v * 3], xv, 12);
            s->indices[v] = v; /* m_vertexBuffer.size() - 1 */
            memcpy(&s->normals[v * 3], xnn, 12);
            memcpy(&s->colors[v * 3], color, 12);
            s->nverts++;
        }
        s->material_ids[s->ntris++] = material_id;
    }
}
/* The five vectors of src/engine/Scene.h:25-29 are PUBLIC members: an application may fill them with anything (the
 * per-corner normals and colours need not be equal).  This appends tri_count unindexed triangles exactly as given --
 * it is how the tests feed smooth-normal / per-vertex-colour meshes, which Raytracing.metal:95-112 interpolates. */
void orc_scene_add_raw(orc_scene *s, const float *pos9, const float *nrm9, const float *col9, const uint32_t *material_ids,
                       int tri_count)
{
    scene_reserve(s, (uint32_t)tri_count);
    if (s->accel) { accel_free(s->accel); s->accel = NULL; }
    for (int i = 0; i < tri_count; ++i) {
        for (int j = 0; j < 3; ++j) {
            uint32_t v = s->nverts;
            memcpy(&s->positions[v * 3], &pos9[(i * 3 + j) * 3], 12);
            memcpy(&s->normals[v * 3], &nrm9[(i * 3 + j) * 3], 12);
            memcpy(&s->colors[v * 3], &col9[(i * 3 + j) * 3], 12);
            s->indices[v] = v;
            s->nverts++;
        }
        s->material_ids[s->ntris++] = material_ids[i];
    }
}
/* src/engine/Scene.cpp:13-22 cubeVertices */
static const float k_cube_verts[8 * 3] = {
    -0.5f, -0.5f, -0.5f, 0.5f, -0.5f, -0.5f, -0.5f, 0.5f, -0.5f, 0.5f, 0.5f, -0.5f,
    -0.5f, -0.5f, 0.5f,  0.5f, -0.5f, 0.5f,  -0.5f, 0.5f, 0.5f,  0.5f, 0.5f, 0.5f,
};
void orc_scene_add_cube(orc_scene *s, const float color[3], const float *mtx)
{
    /* Scene.cpp:24-58 */
    static const uint32_t tris[36] = { 0, 4, 6, 0, 6, 2, 1, 3, 7, 1, 7, 5, 0, 1, 5, 0, 5, 4,
                                       2, 6, 7, 2, 7, 3, 0, 2, 3, 0, 3, 1, 4, 5, 7, 4, 7, 6 };
    orc_scene_add_geometry(s, k_cube_verts, tris, 12, mtx, color, 1u /* MATERIAL_DEFAULT, Renderer.h:16 */);
}
static void plane_verts(float out[12])
{
    /* Scene.cpp:62-67 / 79-84: cubeVertices[0], [1], [5], [4] */
    static const int sel[4] = { 0, 1, 5, 4 };
    for (int i = 0; i < 4; ++i) memcpy(&out[i * 3], &k_cube_verts[sel[i] * 3], 12);
}
void orc_scene_add_plane(orc_scene *s, const float color[3], const float *mtx)
{
    /* Scene.cpp:60-75 */
    float v[12]; plane_verts(v);
    static const uint32_t tris[6] = { 0, 2, 1, 0, 3, 2 };
    orc_scene_add_geometry(s, v, tris, 2, mtx, color, 1u);
}
void orc_scene_add_area_light(orc_scene *s, const float color[3], const float *mtx)
{
    /* Scene.cpp:77-92 */
    float v[12]; plane_verts(v);
    static const uint32_t tris[6] = { 0, 2, 1, 0, 3, 2 };
    orc_scene_add_geometry(s, v, tris, 2, mtx, color, 2u /* MATERIAL_EMISSIVE, Renderer.h:17 */);
}
void orc_scene_cornell_box(orc_scene *s)
{
    /* src/cornellBox.h:11-52 */
    const float kPi = 3.1415926535897932384626433832795f;
    const float white[3] = { 0.725f, 0.71f, 0.68f };
    const float red[3] = { 0.63f, 0.065f, 0.05f };
    const float green[3] = { 0.14f, 0.491f, 0.05f };
    const float one[3] = { 1.0f, 1.0f, 1.0f };
    float m[16];
    orc_mtx_srt(m, 0.6f, 0.6f, 0.6f, 0.0f, 0.3f, 0.0f, 0.3275f, 0.3f, 0.3725f);
    orc_scene_add_cube(s, white, m); /* short box */
    orc_mtx_srt(m, 0.6f, 1.2f, 0.6f, 0.0f, -0.3f, 0.0f, -0.335f, 0.6f, -0.29f);
    orc_scene_add_cube(s, white, m); /* tall box */
    orc_mtx_srt(m, 2.0f, 2.0f, 2.0f, 0.0f, 0.0f, kPi, 0.0f, 1.0f, 0.0f);
    orc_scene_add_plane(s, white, m); /* ceiling */
    orc_mtx_srt(m, 2.0f, 2.0f, 2.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f);
    orc_scene_add_plane(s, white, m); /* floor */
    orc_mtx_srt(m, 2.0f, 2.0f, 2.0f, 0.0f, 0.0f, kPi / 2.0f, 0.0f, 1.0f, 0.0f);
    orc_scene_add_plane(s, red, m); /* left wall */
    orc_mtx_srt(m, 2.0f, 2.0f, 2.0f, 0.0f, 0.0f, -kPi / 2.0f, 0.0f, 1.0f, 0.0f);
    orc_scene_add_plane(s, green, m); /* right wall */
    orc_mtx_srt(m, 2.0f, 2.0f, 2.0f, -kPi / 2.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f);
    orc_scene_add_plane(s, white, m); /* back wall */
    orc_mtx_srt(m, 0.5f, 1.98f, 0.5f, 0.0f, 0.0f, kPi, 0.0f, 1.0f, 0.0f);
    orc_scene_add_area_light(s, one, m); /* light */
}
/* SURVEY 8d C4 (a synthetic this project defines, not in the reference): Cornell box + the
 * reference cube mesh replicated on an n x n x n lattice inside the room.  Lattice region
 * x,z in [-0.9,0.9], y in [0.05,1.85]; cube side = 0.25 * cell; per-cube jitter and y-rotation
 * from pcg(seed).  n = 44 -> 36 + 12*44^3 = 1,022,244 triangles. */
void orc_scene_cornell_lattice(orc_scene *s, int n, uint32_t seed)
{
    orc_scene_cornell_box(s);
    const float lo[3] = { -0.9f, 0.05f, -0.9f }, hi[3] = { 0.9f, 1.85f, 0.9f };
    float cell[3];
    for (int a = 0; a < 3; ++a) cell[a] = (hi[a] - lo[a]) / (float)n;
    uint32_t id = 0;
    for (int iz = 0; iz < n; ++iz)
        for (int iy = 0; iy < n; ++iy)
            for (int ix = 0; ix < n; ++ix, ++id) {
                uint32_t h0 = orc_pcg_hash32(seed ^ (id * 4u + 0u));
                uint32_t h1 = orc_pcg_hash32(seed ^ (id * 4u + 1u));
                uint32_t h2 = orc_pcg_hash32(seed ^ (id * 4u + 2u));
                uint32_t h3 = orc_pcg_hash32(seed ^ (id * 4u + 3u));
                float j0 = (float)(h0 >> 8) * (1.0f / 16777216.0f) - 0.5f;
                float j1 = (float)(h1 >> 8) * (1.0f / 16777216.0f) - 0.5f;
                float j2 = (float)(h2 >> 8) * (1.0f / 16777216.0f) - 0.5f;
                float rot = (float)(h3 >> 8) * (1.0f / 16777216.0f) * 1.5707964f;
                float cx = lo[0] + ((float)ix + 0.5f + 0.3f * j0) * cell[0];
                float cy = lo[1] + ((float)iy + 0.5f + 0.3f * j1) * cell[1];
                float cz = lo[2] + ((float)iz + 0.5f + 0.3f * j2) * cell[2];
                float side = 0.25f * fminf(cell[0], fminf(cell[1], cell[2]));
                float col[3] = { 0.3f + 0.6f * (float)((h0 >> 3) & 31) / 31.0f,
                                 0.3f + 0.6f * (float)((h1 >> 3) & 31) / 31.0f,
                                 0.3f + 0.6f * (float)((h2 >> 3) & 31) / 31.0f };
                float m[16];
                orc_mtx_srt(m, side, side, side, 0.0f, rot, 0.0f, cx, cy, cz);
                orc_scene_add_cube(s, col, m);
            }
}

/* =============================== camera + uniforms (a2, a3) =============================== */
void orc_make_uniforms(orc_uniforms *u, uint32_t w, uint32_t h, uint32_t frame_index,
                       const float eye[3], const float at[3])
{
    memset(u, 0, sizeof(*u));
    /* Renderer.cpp:18-27 aspect; Renderer.cpp:84-89 updateCamera; up = (0,1,0) Renderer.cpp:7 */
    float aspect = (float)(int)w / (float)(int)h;
    const float up[3] = { 0.0f, 1.0f, 0.0f };
    float view[16], proj[16], view_proj[16], inv[16];
    orc_mtx_look_at_rh(view, eye, at, up);
    orc_mtx_proj_rh(proj, 45.0f, aspect, 1.0f, 125.0f);
    orc_mtx_mul(view_proj, view, proj);
    /* MetalRenderer.mm:346-353 */
    orc_mtx_inverse(inv, view_proj);
    u->cam_pos[0] = eye[0]; u->cam_pos[1] = eye[1]; u->cam_pos[2] = eye[2];
    /* MetalUniforms.h:49-59: data.columns[j][i] = in[i*4+j] */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) u->inv_view_proj[j * 4 + i] = inv[i * 4 + j];
    /* MetalRenderer.mm:355-359 */
    u->light_pos[0] = 0.0f; u->light_pos[1] = 1.98f; u->light_pos[2] = 0.0f;
    u->light_forward[0] = 0.0f; u->light_forward[1] = -1.0f; u->light_forward[2] = 0.0f;
    u->light_right[0] = 0.25f; u->light_right[1] = 0.0f; u->light_right[2] = 0.0f;
    u->light_up[0] = 0.0f; u->light_up[1] = 0.0f; u->light_up[2] = 0.25f;
    u->light_color[0] = 1.0f; u->light_color[1] = 1.0f; u->light_color[2] = 1.0f;
    /* MetalRenderer.mm:361-363 */
    u->width = w; u->height = h; u->frameIndex = frame_index;
}

/* =============================== pixel offsets (a4) =============================== */
uint32_t orc_pcg_hash32(uint32_t v)
{
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
void orc_pixel_offsets(uint32_t seed, uint32_t w, uint32_t h, uint32_t *out)
{
    /* replaces the unseeded libc rand() bytes of Texture.cpp:16-29 */
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) out[(size_t)y * w + x] = orc_pcg_hash32(seed ^ (y * w + x));
}

/* =============================== sampling (a5, a10, a11) =============================== */
static const uint32_t k_primes[ORC_HALTON_DIMS] = {
    /* common.h:53-59 (first 16), extended with the following primes */
    2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53,
    59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113, 127, 131,
    137, 139, 149, 151, 157, 163, 167, 173, 179, 181, 191, 193, 197, 199, 211, 223,
    227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307, 311,
};
uint32_t orc_halton_prime(uint32_t d) { return k_primes[d % ORC_HALTON_DIMS]; }
#ifdef ORC_TUNED
/* the tuned build (bench.py's CPU-baseline leg only): the SAME recurrence with the base as a compile-time constant per dimension, so that
 * i % b and i / b become a multiply and a shift -- identical integers, identical floats, about five times faster */
static inline __attribute__((always_inline)) float halton_const_base(uint32_t i, const uint32_t b)
{
    float f = 1.0f, invB = 1.0f / b, r = 0;
    while (i > 0) { f = f * invB; r = r + f * (i % b); i = i / b; }
    return r;
}
#endif
float orc_halton(uint32_t i, uint32_t d)
{
#ifdef ORC_TUNED
    switch (d) {
    case 0: return halton_const_base(i, 2u);
    case 1: return halton_const_base(i, 3u);
    case 2: return halton_const_base(i, 5u);
    case 3: return halton_const_base(i, 7u);
    case 4: return halton_const_base(i, 11u);
    case 5: return halton_const_base(i, 13u);
    case 6: return halton_const_base(i, 17u);
    case 7: return halton_const_base(i, 19u);
    case 8: return halton_const_base(i, 23u);
    case 9: return halton_const_base(i, 29u);
    case 10: return halton_const_base(i, 31u);
    case 11: return halton_const_base(i, 37u);
    case 12: return halton_const_base(i, 41u);
    case 13: return halton_const_base(i, 43u);
    case 14: return halton_const_base(i, 47u);
    case 15: return halton_const_base(i, 53u);
    case 16: return halton_const_base(i, 59u);
    case 17: return halton_const_base(i, 61u);
    case 18: return halton_const_base(i, 67u);
    case 19: return halton_const_base(i, 71u);
    case 20: return halton_const_base(i, 73u);
    case 21: return halton_const_base(i, 79u);
    case 22: return halton_const_base(i, 83u);
    case 23: return halton_const_base(i, 89u);
    case 24: return halton_const_base(i, 97u);
    case 25: return halton_const_base(i, 101u);
    case 26: return halton_const_base(i, 103u);
    case 27: return halton_const_base(i, 107u);
    case 28: return halton_const_base(i, 109u);
    case 29: return halton_const_base(i, 113u);
    case 30: return halton_const_base(i, 127u);
    case 31: return halton_const_base(i, 131u);
    case 32: return halton_const_base(i, 137u);
    case 33: return halton_const_base(i, 139u);
    case 34: return halton_const_base(i, 149u);
    case 35: return halton_const_base(i, 151u);
    case 36: return halton_const_base(i, 157u);
    case 37: return halton_const_base(i, 163u);
    case 38: return halton_const_base(i, 167u);
    case 39: return halton_const_base(i, 173u);
    case 40: return halton_const_base(i, 179u);
    case 41: return halton_const_base(i, 181u);
    case 42: return halton_const_base(i, 191u);
    case 43: return halton_const_base(i, 193u);
    case 44: return halton_const_base(i, 197u);
    case 45: return halton_const_base(i, 199u);
    case 46: return halton_const_base(i, 211u);
    case 47: return halton_const_base(i, 223u);
    case 48: return halton_const_base(i, 227u);
    case 49: return halton_const_base(i, 229u);
    case 50: return halton_const_base(i, 233u);
    case 51: return halton_const_base(i, 239u);
    case 52: return halton_const_base(i, 241u);
    case 53: return halton_const_base(i, 251u);
    case 54: return halton_const_base(i, 257u);
    case 55: return halton_const_base(i, 263u);
    case 56: return halton_const_base(i, 269u);
    case 57: return halton_const_base(i, 271u);
    case 58: return halton_const_base(i, 277u);
    case 59: return halton_const_base(i, 281u);
    case 60: return halton_const_base(i, 283u);
    case 61: return halton_const_base(i, 293u);
    case 62: return halton_const_base(i, 307u);
    case 63: return halton_const_base(i, 311u);
    default: break;
    }
#endif
    /* common.h:51-75 */
    uint32_t b = k_primes[d];
    float f = 1.0f;
    float invB = 1.0f / b;
    float r = 0;
    while (i > 0) {
        f = f * invB;
        r = r + f * (i % b);
        i = i / b;
    }
    return r;
}
void orc_sample_cosine_hemisphere(const float u[2], float out[3])
{
    /* common.h:80-91.  PI (common.h:9) is an unsuffixed literal = fp32 in MSL. */
    const float PI_F = 3.1415926535898f;
    float phi = 2.0f * PI_F * u[0];
    float cos_phi, sin_phi;
    if (g_trig_mode == ORC_TRIG_PORTABLE) {
        orc_sincos_portable(phi, &sin_phi, &cos_phi);
    } else {
        cos_phi = cosf(phi);
        sin_phi = sinf(phi);
    }
    float cos_theta = sqrtf(u[1]);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    out[0] = sin_theta * cos_phi;
    out[1] = cos_theta;
    out[2] = sin_theta * sin_phi;
}
void orc_align_hemisphere(const float sample[3], const float normal[3], float out[3])
{
    /* common.h:95-110 */
    const float up[3] = { normal[0], normal[1], normal[2] };
    const float k[3] = { 0.0072f, 1.0f, 0.0034f };
    float c[3]; cross3(normal, k, c);
    float right[3]; normalize3(c, right);
    float forward[3]; cross3(right, up, forward);
    /* sample.x * right + sample.y * up + sample.z * forward, left to right */
    for (int a = 0; a < 3; ++a) out[a] = sample[0] * right[a] + sample[1] * up[a] + sample[2] * forward[a];
}
void orc_sample_area_light(const orc_uniforms *u, const float r[2], const float pos[3],
                           const float nrm[3], float dir[3], float color[3], float *dist)
{
    /* common.h:119-159 */
    float ux = r[0] * 2.0f - 1.0f;
    float uy = r[1] * 2.0f - 1.0f;
    float sp[3];
    for (int a = 0; a < 3; ++a) sp[a] = u->light_pos[a] + u->light_right[a] * ux + u->light_up[a] * uy;
    for (int a = 0; a < 3; ++a) dir[a] = sp[a] - pos[a];
    float d = sqrtf(dot3(dir, dir)); /* length() */
    float inv = 1.0f / fmaxf(d, 1e-3f);
    for (int a = 0; a < 3; ++a) dir[a] *= inv;
    float c[3] = { u->light_color[0], u->light_color[1], u->light_color[2] };
    float inv2 = inv * inv;
    for (int a = 0; a < 3; ++a) c[a] *= inv2;
    float nd[3] = { -dir[0], -dir[1], -dir[2] };
    float k1 = saturatef(dot3(nd, u->light_forward));
    for (int a = 0; a < 3; ++a) c[a] *= k1;
    float k2 = saturatef(dot3(nrm, dir));
    for (int a = 0; a < 3; ++a) c[a] *= k2;
    if (g_variant & ORC_VAR_LIGHT_NORMALISED) {   /* negative control only: area / pi, area of the quad = |2 right x 2 up| */
        float cr[3] = { u->light_right[1] * u->light_up[2] - u->light_right[2] * u->light_up[1],
                        u->light_right[2] * u->light_up[0] - u->light_right[0] * u->light_up[2],
                        u->light_right[0] * u->light_up[1] - u->light_right[1] * u->light_up[0] };
        float k3 = 4.0f * sqrtf(dot3(cr, cr)) / 3.14159265f;
        for (int a = 0; a < 3; ++a) c[a] *= k3;
    }
    color[0] = c[0]; color[1] = c[1]; color[2] = c[2];
    *dist = d;
}

/* =============================== kernels =============================== */
void orc_raygen(const orc_uniforms *u, const uint32_t *offsets, orc_ray *rays, float *dst, uint32_t x, uint32_t y)
{
    /* Raytracing.metal:41-91 */
    if (!(x < u->width && y < u->height)) return;
    uint32_t rayIdx = y * u->width + x;
    orc_ray *ray = &rays[rayIdx];
    float px = (float)x, py = (float)y;
    uint32_t offset = offsets[rayIdx];
    float r0 = orc_halton(offset + u->frameIndex, 0);
    float r1 = orc_halton(offset + u->frameIndex, 1);
    px += r0; py += r1;
    float uvx = px / (float)u->width, uvy = py / (float)u->height;
    uvx = uvx * 2.0f - 1.0f; uvy = uvy * 2.0f - 1.0f;
    /* float4(uv, 0, 1) * M  (row vector times matrix): world[j] = dot(v, M.columns[j]) */
    const float *m = u->inv_view_proj;
    float v[4] = { uvx, uvy, 0.0f, 1.0f }, world[4];
    for (int j = 0; j < 4; ++j)
        world[j] = v[0] * m[j * 4 + 0] + v[1] * m[j * 4 + 1] + v[2] * m[j * 4 + 2] + v[3] * m[j * 4 + 3];
    world[0] /= world[3]; world[1] /= world[3]; world[2] /= world[3];
    float d[3] = { world[0] - u->cam_pos[0], world[1] - u->cam_pos[1], world[2] - u->cam_pos[2] };
    float dn[3]; normalize3(d, dn);
    ray->origin[0] = u->cam_pos[0]; ray->origin[1] = u->cam_pos[1]; ray->origin[2] = u->cam_pos[2];
    ray->direction[0] = dn[0]; ray->direction[1] = dn[1]; ray->direction[2] = dn[2];
    ray->mask = 3u; /* RAY_MASK_PRIMARY, Raytracing.metal:17 */
    ray->maxDistance = INFINITY;
    ray->color[0] = 1.0f; ray->color[1] = 1.0f; ray->color[2] = 1.0f; ray->color[3] = 0.0f;
    float *o = &dst[(size_t)rayIdx * 4];
    o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f;
}

/* Raytracing.metal:95-112 */
static void interp_attr(const float *attr, const orc_isect *is, float out[3])
{
    float ux = is->coordinates[0], uy = is->coordinates[1];
    float uz = 1.0f - ux - uy;
    uint32_t t = (uint32_t)is->primitiveIndex;
    const float *T0 = &attr[(t * 3 + 0) * 3], *T1 = &attr[(t * 3 + 1) * 3], *T2 = &attr[(t * 3 + 2) * 3];
    for (int a = 0; a < 3; ++a) out[a] = ux * T0[a] + uy * T1[a] + uz * T2[a];
}

/* Albedo texel of a shaded hit: project definition (orc_scene_set_textures).  Texture coordinates are interpolated with the
 * weights of Raytracing.metal:95-112; nearest texel, repeat wrap, row 0 of the image at v = 0; texel RGB / 255 multiplies vc. */
static void texture_albedo(const orc_scene *s, const orc_isect *is, float vc[3])
{
    uint32_t t = (uint32_t)is->primitiveIndex;
    uint32_t id = s->tex_ids[t];
    if (id == 0u) return;
    float ux = is->coordinates[0], uy = is->coordinates[1];
    float uz = 1.0f - ux - uy;
    const float *p = &s->uvs[(size_t)t * 6];
    float uu = ux * p[0] + uy * p[2] + uz * p[4];
    float vv = ux * p[1] + uy * p[3] + uz * p[5];
    const uint32_t *tab = &s->tex_table[(id - 1u) * 4];
    uint32_t w = tab[1], h = tab[2];
    float fu = uu - floorf(uu), fv = vv - floorf(vv);
    uint32_t px = (uint32_t)(fu * (float)w), py = (uint32_t)(fv * (float)h);
    if (px >= w) px = w - 1u;
    if (py >= h) py = h - 1u;
    uint32_t texel = s->texels[tab[0] + py * w + px];
    vc[0] = vc[0] * ((float)(texel & 255u) / 255.0f);
    vc[1] = vc[1] * ((float)((texel >> 8) & 255u) / 255.0f);
    vc[2] = vc[2] * ((float)((texel >> 16) & 255u) / 255.0f);
}

void orc_primary_hit(const orc_uniforms *u, orc_ray *rays, orc_ray *shadow_rays, const orc_isect *isects,
                     const orc_scene *s, uint32_t bounce, const uint32_t *offsets, float *dst,
                     uint32_t x, uint32_t y, orc_stats *st)
{
    /* Raytracing.metal:115-215 */
    if (x >= u->width || y >= u->height) return;
    uint32_t rayIdx = y * u->width + x;
    orc_ray *ray = &rays[rayIdx];
    orc_ray *shadowRay = &shadow_rays[rayIdx];
    const orc_isect *is = &isects[rayIdx];
    if (ray->maxDistance < 0.0f || is->distance < 0.0f) {
        ray->maxDistance = -1.0f;
        shadowRay->maxDistance = -1.0f;
        return;
    }
    float color[3] = { ray->color[0], ray->color[1], ray->color[2] };
    uint32_t materialID = s->material_ids[is->primitiveIndex];
    if (materialID == 1u) {
        float P[3];
        for (int a = 0; a < 3; ++a) P[a] = ray->origin[a] + ray->direction[a] * is->distance;
        float vc[3], vn0[3], vn[3];
        interp_attr(s->colors, is, vc);
        if (s->uvs) texture_albedo(s, is, vc);
        interp_attr(s->normals, is, vn0);
        normalize3(vn0, vn);
        uint32_t offset = offsets[rayIdx];
        float r[2] = { orc_halton(offset + u->frameIndex, 2 + bounce * 4 + 0),
                       orc_halton(offset + u->frameIndex, 2 + bounce * 4 + 1) };
        float ldir[3], lcol[3], ldist;
        orc_sample_area_light(u, r, P, vn, ldir, lcol, &ldist);
        for (int a = 0; a < 3; ++a) color[a] *= vc[a];
        for (int a = 0; a < 3; ++a) shadowRay->origin[a] = P[a] + vn[a] * 1e-3f;
        for (int a = 0; a < 3; ++a) shadowRay->direction[a] = ldir[a];
        shadowRay->mask = 1u; /* RAY_MASK_SHADOW */
        shadowRay->maxDistance = ldist - 1e-3f;
        for (int a = 0; a < 3; ++a) shadowRay->color[a] = lcol[a] * color[a];
        r[0] = orc_halton(offset + u->frameIndex, 2 + bounce * 4 + 2);
        r[1] = orc_halton(offset + u->frameIndex, 2 + bounce * 4 + 3);
        float sd[3], ad[3];
        orc_sample_cosine_hemisphere(r, sd);
        orc_align_hemisphere(sd, vn, ad);
        for (int a = 0; a < 3; ++a) ray->origin[a] = P[a] + vn[a] * 1e-3f;
        for (int a = 0; a < 3; ++a) ray->direction[a] = ad[a];
        for (int a = 0; a < 3; ++a) ray->color[a] = color[a];
        ray->mask = (g_variant & ORC_VAR_BOUNCE_SEES_LIGHT) ? 3u : 1u; /* RAY_MASK_SECONDARY (the variant: negative control only) */
        if (st) st->shaded_hits++;
    } else if (materialID == 2u) {
        float *o = &dst[(size_t)rayIdx * 4];
        o[0] = u->light_color[0]; o[1] = u->light_color[1]; o[2] = u->light_color[2]; o[3] = 1.0f;
        ray->maxDistance = -1.0f;
        shadowRay->maxDistance = -1.0f;
    } else {
        float *o = &dst[(size_t)rayIdx * 4];
        o[0] = 1.0f; o[1] = 0.0f; o[2] = 1.0f; o[3] = 1.0f;
        /* The reference leaves ray and shadowRay untouched here (Raytracing.metal:210-214), so the
         * shadow ray is whatever the previous bounce/frame left behind (uninitialised private
         * memory on the first frame).  Project rule: the shadow ray is disabled. */
        shadowRay->maxDistance = -1.0f;
    }
}

void orc_shadow_hit(const orc_uniforms *u, const orc_ray *shadow_rays, const float *shadow_dist,
                    const float *src, float *dst, uint32_t x, uint32_t y)
{
    /* Raytracing.metal:219-246 */
    if (!(x < u->width && y < u->height)) return;
    uint32_t rayIdx = y * u->width + x;
    const orc_ray *sr = &shadow_rays[rayIdx];
    float d = shadow_dist[rayIdx];
    float c[3] = { src[(size_t)rayIdx * 4 + 0], src[(size_t)rayIdx * 4 + 1], src[(size_t)rayIdx * 4 + 2] };
    if (sr->maxDistance >= 0.0f && d < 0.0f) {
        c[0] += sr->color[0]; c[1] += sr->color[1]; c[2] += sr->color[2];
    }
    float *o = &dst[(size_t)rayIdx * 4];
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = 1.0f;
}

void orc_accumulate(const orc_uniforms *u, const float *render, const float *prev, float *accum,
                    uint32_t x, uint32_t y)
{
    /* Accumulate.metal:19-39 */
    if (!(x < u->width && y < u->height)) return;
    size_t p = ((size_t)y * u->width + x) * 4;
    float c[3] = { render[p], render[p + 1], render[p + 2] };
    if (u->frameIndex > 0) {
        float pc[3] = { prev[p], prev[p + 1], prev[p + 2] };
        float f = (float)u->frameIndex;
        for (int a = 0; a < 3; ++a) pc[a] *= f;
        for (int a = 0; a < 3; ++a) c[a] += pc[a];
        float f1 = (float)(u->frameIndex + 1);
        for (int a = 0; a < 3; ++a) c[a] /= f1;
    }
    accum[p] = c[0]; accum[p + 1] = c[1]; accum[p + 2] = c[2]; accum[p + 3] = 1.0f;
}

/* =============================== intersector (a7, a12) =============================== */
/* INTERSECTION CONTRACT (project definition; the reference's intersector is closed source):
 *  - triangle k has vertices positions[indices[3k..3k+2]] and mask = material_ids[k]
 *    (MetalRenderer.mm:274-277); it is a candidate iff (ray.mask & mask) != 0
 *    (MPSRayMaskOptionPrimitive, MetalRenderer.mm:269);
 *  - two-sided Moeller-Trumbore on (v0, e1 = v1-v0, e2 = v2-v0) in fp32, no epsilon:
 *      p = d x e2; det = e1.p; inv = 1/det; tv = o - v0; u = (tv.p)*inv; q = tv x e1;
 *      v = (d.q)*inv; t = (e2.q)*inv;  hit iff u>=0, v>=0, u+v<=1, 0<=t<=maxDistance;
 *  - nearest: minimum t; equal t -> lower primitive index; coordinates = (1-u-v, u)
 *    (weights of vertex 0 and 1, Raytracing.metal:99-111);
 *  - rays with maxDistance < 0 are inactive -> distance = -1 (MetalRenderer.mm / MPS semantics);
 *  - miss -> distance = -1, primitiveIndex = -1 (project choice; the reference only tests "< 0").
 */
typedef struct { float v0[3], e1[3], e2[3]; } tri9;

static inline int tri_test(const float o[3], const float d[3], const float *v0, const float *e1, const float *e2,
                           float *t_out, float *u_out, float *v_out)
{
    float p[3]; cross3(d, e2, p);
    float det = dot3(e1, p);
    float inv = 1.0f / det;
    float tv[3] = { o[0] - v0[0], o[1] - v0[1], o[2] - v0[2] };
    float u = dot3(tv, p) * inv;
    float q[3]; cross3(tv, e1, q);
    float v = dot3(d, q) * inv;
    float t = dot3(e2, q) * inv;
    if (!(u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f)) return 0;
    *t_out = t; *u_out = u; *v_out = v;
    return 1;
}
static inline void tri_fetch(const orc_scene *s, uint32_t k, float v0[3], float e1[3], float e2[3])
{
    const float *a = &s->positions[s->indices[k * 3 + 0] * 3];
    const float *b = &s->positions[s->indices[k * 3 + 1] * 3];
    const float *c = &s->positions[s->indices[k * 3 + 2] * 3];
    for (int i = 0; i < 3; ++i) { v0[i] = a[i]; e1[i] = b[i] - a[i]; e2[i] = c[i] - a[i]; }
}

static void nearest_brute_one(const orc_scene *s, const orc_ray *r, orc_isect *out, orc_stats *st)
{
    out->distance = -1.0f; out->primitiveIndex = -1; out->coordinates[0] = 0.0f; out->coordinates[1] = 0.0f;
    if (r->maxDistance < 0.0f) return;
    float best = r->maxDistance;
    int found = 0;
    for (uint32_t k = 0; k < s->ntris; ++k) {
        if ((r->mask & s->material_ids[k]) == 0) continue;
        float v0[3], e1[3], e2[3], t, u, v;
        tri_fetch(s, k, v0, e1, e2);
        if (st) st->tri_tests++;
        if (!tri_test(r->origin, r->direction, v0, e1, e2, &t, &u, &v)) continue;
        if (!(t >= 0.0f && t <= r->maxDistance)) continue;
        if (!found || t < best) { /* ascending k: ties keep the lower index */
            found = 1; best = t;
            out->distance = t; out->primitiveIndex = (int32_t)k;
            out->coordinates[0] = 1.0f - u - v; out->coordinates[1] = u;
        }
    }
}
static float any_brute_one(const orc_scene *s, const orc_ray *r, orc_stats *st)
{
    if (r->maxDistance < 0.0f) return -1.0f;
    for (uint32_t k = 0; k < s->ntris; ++k) {
        if ((r->mask & s->material_ids[k]) == 0) continue;
        float v0[3], e1[3], e2[3], t, u, v;
        tri_fetch(s, k, v0, e1, e2);
        if (st) st->tri_tests++;
        if (!tri_test(r->origin, r->direction, v0, e1, e2, &t, &u, &v)) continue;
        if (t >= 0.0f && t <= r->maxDistance) return t;
    }
    return -1.0f;
}
void orc_intersect_nearest_brute(const orc_scene *s, const orc_ray *rays, size_t n, orc_isect *out)
{
    for (size_t i = 0; i < n; ++i) nearest_brute_one(s, &rays[i], &out[i], NULL);
}
void orc_intersect_any_brute(const orc_scene *s, const orc_ray *rays, size_t n, float *out)
{
    for (size_t i = 0; i < n; ++i) out[i] = any_brute_one(s, &rays[i], NULL);
}

/* ---- oracle BVH for large scenes: median split on the longest centroid axis, leaves <= 4 tris,
 *      boxes padded so that every triangle the brute-force test accepts is also reached. ---- */
typedef struct { float lo[3], hi[3]; int32_t left, right; uint32_t first, count; } onode;
typedef struct { onode *nodes; uint32_t nnodes; uint32_t *order; float *cent; } oaccel;

static void accel_free(void *a_)
{
    oaccel *a = (oaccel *)a_;
    if (!a) return;
    free(a->nodes); free(a->order); free(a->cent); free(a);
}
static void tri_bounds(const orc_scene *s, uint32_t k, float lo[3], float hi[3])
{
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int j = 0; j < 3; ++j) {
        const float *p = &s->positions[s->indices[k * 3 + j] * 3];
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
    }
}
static int g_sort_axis; static const float *g_sort_cent;
static int cmp_cent(const void *x, const void *y)
{
    float a = g_sort_cent[*(const uint32_t *)x * 3 + g_sort_axis], b = g_sort_cent[*(const uint32_t *)y * 3 + g_sort_axis];
    if (a < b) return -1; if (a > b) return 1;
    uint32_t ia = *(const uint32_t *)x, ib = *(const uint32_t *)y;
    return ia < ib ? -1 : (ia > ib ? 1 : 0);
}
static uint32_t build_rec(const orc_scene *s, oaccel *A, uint32_t first, uint32_t count, float pad)
{
    uint32_t me = A->nnodes++;
    onode *n = &A->nodes[me];
    float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int a = 0; a < 3; ++a) { n->lo[a] = INFINITY; n->hi[a] = -INFINITY; }
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t k = A->order[first + i];
        float lo[3], hi[3]; tri_bounds(s, k, lo, hi);
        for (int a = 0; a < 3; ++a) {
            n->lo[a] = fminf(n->lo[a], lo[a]); n->hi[a] = fmaxf(n->hi[a], hi[a]);
            clo[a] = fminf(clo[a], A->cent[k * 3 + a]); chi[a] = fmaxf(chi[a], A->cent[k * 3 + a]);
        }
    }
    for (int a = 0; a < 3; ++a) { n->lo[a] -= pad; n->hi[a] += pad; }
    n->first = first; n->count = count; n->left = n->right = -1;
    if (count <= 4) return me;
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    g_sort_axis = axis; g_sort_cent = A->cent;
    qsort(&A->order[first], count, sizeof(uint32_t), cmp_cent);
    uint32_t half = count / 2;
    uint32_t l = build_rec(s, A, first, half, pad);
    uint32_t r = build_rec(s, A, first + half, count - half, pad);
    A->nodes[me].left = (int32_t)l; A->nodes[me].right = (int32_t)r; A->nodes[me].count = 0;
    return me;
}
static oaccel *accel_get(orc_scene *s)
{
    if (s->accel) return (oaccel *)s->accel;
    oaccel *A = (oaccel *)calloc(1, sizeof(oaccel));
    A->nodes = (onode *)malloc(sizeof(onode) * (2 * (size_t)s->ntris + 1));
    A->order = (uint32_t *)malloc(sizeof(uint32_t) * s->ntris);
    A->cent = (float *)malloc(sizeof(float) * 3 * s->ntris);
    float slo[3] = { INFINITY, INFINITY, INFINITY }, shi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t k = 0; k < s->ntris; ++k) {
        A->order[k] = k;
        float lo[3], hi[3]; tri_bounds(s, k, lo, hi);
        for (int a = 0; a < 3; ++a) {
            A->cent[k * 3 + a] = 0.5f * (lo[a] + hi[a]);
            slo[a] = fminf(slo[a], lo[a]); shi[a] = fmaxf(shi[a], hi[a]);
        }
    }
    float diag = fmaxf(shi[0] - slo[0], fmaxf(shi[1] - slo[1], shi[2] - slo[2]));
    float pad = 1e-4f * fmaxf(diag, 1e-6f);
    build_rec(s, A, 0, s->ntris, pad);
    s->accel = A;
    return A;
}
static inline int box_hit(const onode *n, const float o[3], const float inv[3], float tmax)
{
    float t0 = 0.0f, t1 = tmax;
    for (int a = 0; a < 3; ++a) {
        float ta = (n->lo[a] - o[a]) * inv[a], tb = (n->hi[a] - o[a]) * inv[a];
        float tn = fminf(ta, tb), tf = fmaxf(ta, tb); /* fmin/fmax drop NaN (0*inf) */
        t0 = fmaxf(t0, tn); t1 = fminf(t1, tf);
    }
    /* generous slack: the box test only has to be conservative */
    return t0 <= t1 * 1.00001f + 1e-6f;
}
static void nearest_bvh_one(const orc_scene *s, const oaccel *A, const orc_ray *r, orc_isect *out, orc_stats *st)
{
    out->distance = -1.0f; out->primitiveIndex = -1; out->coordinates[0] = 0.0f; out->coordinates[1] = 0.0f;
    if (r->maxDistance < 0.0f) return;
    float inv[3] = { 1.0f / r->direction[0], 1.0f / r->direction[1], 1.0f / r->direction[2] };
    float best = r->maxDistance; int found = 0; int32_t best_k = -1;
    uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const onode *n = &A->nodes[stack[--sp]];
        if (st) st->node_visits++;
        if (!box_hit(n, r->origin, inv, best)) continue;
        if (n->left < 0) {
            for (uint32_t i = 0; i < n->count; ++i) {
                uint32_t k = A->order[n->first + i];
                if ((r->mask & s->material_ids[k]) == 0) continue;
                float v0[3], e1[3], e2[3], t, u, v;
                tri_fetch(s, k, v0, e1, e2);
                if (st) st->tri_tests++;
                if (!tri_test(r->origin, r->direction, v0, e1, e2, &t, &u, &v)) continue;
                if (!(t >= 0.0f && t <= r->maxDistance)) continue;
                if (!found || t < best || (t == best && (int32_t)k < best_k)) {
                    found = 1; best = t; best_k = (int32_t)k;
                    out->distance = t; out->primitiveIndex = (int32_t)k;
                    out->coordinates[0] = 1.0f - u - v; out->coordinates[1] = u;
                }
            }
        } else {
            stack[sp++] = (uint32_t)n->left; stack[sp++] = (uint32_t)n->right;
        }
    }
}
static float any_bvh_one(const orc_scene *s, const oaccel *A, const orc_ray *r, orc_stats *st)
{
    if (r->maxDistance < 0.0f) return -1.0f;
    float inv[3] = { 1.0f / r->direction[0], 1.0f / r->direction[1], 1.0f / r->direction[2] };
    uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const onode *n = &A->nodes[stack[--sp]];
        if (st) st->node_visits++;
        if (!box_hit(n, r->origin, inv, r->maxDistance)) continue;
        if (n->left < 0) {
            for (uint32_t i = 0; i < n->count; ++i) {
                uint32_t k = A->order[n->first + i];
                if ((r->mask & s->material_ids[k]) == 0) continue;
                float v0[3], e1[3], e2[3], t, u, v;
                tri_fetch(s, k, v0, e1, e2);
                if (st) st->tri_tests++;
                if (!tri_test(r->origin, r->direction, v0, e1, e2, &t, &u, &v)) continue;
                if (t >= 0.0f && t <= r->maxDistance) return t;
            }
        } else {
            stack[sp++] = (uint32_t)n->left; stack[sp++] = (uint32_t)n->right;
        }
    }
    return -1.0f;
}
/* ---- an acceleration structure from outside (orc_scene_set_bvh): the product's BVH2, near child first, the far one on a stack ---- */
void orc_scene_set_bvh(orc_scene *s, const float *nodes16, uint32_t n_nodes, const float *tris12, uint32_t n_recs)
{
    free(s->xnodes); free(s->xtris);
    s->xnodes = NULL; s->xtris = NULL; s->xn_nodes = 0; s->xn_recs = 0;
    if (!n_nodes || !nodes16 || !tris12) return;
    s->xnodes = (float *)malloc(sizeof(float) * 16 * (size_t)n_nodes);
    s->xtris = (float *)malloc(sizeof(float) * 12 * (size_t)n_recs);
    memcpy(s->xnodes, nodes16, sizeof(float) * 16 * (size_t)n_nodes);
    memcpy(s->xtris, tris12, sizeof(float) * 12 * (size_t)n_recs);
    s->xn_nodes = n_nodes; s->xn_recs = n_recs;
}
static inline float slab_entry(const float *n, int child, const float o[3], const float inv[3], float tmax, int *hit)
{
    /* child 0: x (n0,n1) y (n2,n3) z (n8,n9); child 1: x (n4,n5) y (n6,n7) z (n10,n11) */
    const float lo[3] = { n[child * 4 + 0], n[child * 4 + 2], n[8 + child * 2] }, hi[3] = { n[child * 4 + 1], n[child * 4 + 3], n[9 + child * 2] };
    float t0 = 0.0f, t1 = tmax;
    for (int a = 0; a < 3; ++a) {   /* (comparisons instead of fminf / fmaxf calls: the reciprocals are finite -- clamped away from zero -- so no NaN arises) */
        const float ta = (lo[a] - o[a]) * inv[a], tb = (hi[a] - o[a]) * inv[a];
        const float tn = ta < tb ? ta : tb, tf = ta < tb ? tb : ta;
        t0 = tn > t0 ? tn : t0; t1 = tf < t1 ? tf : t1;
    }
    *hit = t0 <= t1 * 1.00001f + 1e-6f;   /* the product pads its boxes; the slack keeps this walk conservative whatever the contraction */
    return t0;
}
/* any != 0: first hit ends the walk.  Returns 1 when something was hit (out filled for the nearest query). */
static int ext_walk_one(const orc_scene *s, const orc_ray *r, int any, orc_isect *out, orc_stats *st)
{
    if (out) { out->distance = -1.0f; out->primitiveIndex = -1; out->coordinates[0] = 0.0f; out->coordinates[1] = 0.0f; }
    if (r->maxDistance < 0.0f) return 0;
    float inv[3];
    for (int a = 0; a < 3; ++a) {
        float d = r->direction[a];
        if (fabsf(d) < 1e-30f) d = copysignf(1e-30f, d);
        inv[a] = 1.0f / d;
    }
    float best = r->maxDistance; int found = 0; int32_t best_k = -1;
    int32_t stack[128]; int sp = 0;
    int32_t node = 0;
    for (;;) {
        if (node >= 0) {
            const float *n = &s->xnodes[(size_t)node * 16];
            if (st) st->node_visits++;
            int h0, h1;
            const float t0 = slab_entry(n, 0, r->origin, inv, best, &h0), t1 = slab_entry(n, 1, r->origin, inv, best, &h1);
            int32_t c0, c1; memcpy(&c0, &n[12], 4); memcpy(&c1, &n[13], 4);
            if (h0 && h1) {
                const int first1 = t1 < t0;
                stack[sp++] = first1 ? c0 : c1;
                node = first1 ? c1 : c0;
                continue;
            }
            if (h0) { node = c0; continue; }
            if (h1) { node = c1; continue; }
        } else {
            const uint32_t code = ~(uint32_t)node, first = code >> 3, count = (code & 7u) == 7u ? 2u : (code & 7u) + 1u;   /* 7: a quad leaf = two triangles (the product's bvh_build.h) */
            for (uint32_t i = 0; i < count; ++i) {
                const float *rec = &s->xtris[(size_t)(first + i) * 12];
                uint32_t k, mask; memcpy(&k, &rec[3], 4); memcpy(&mask, &rec[7], 4);
                if ((r->mask & mask) == 0) continue;
                float t, u, v;
                if (st) st->tri_tests++;
                if (!tri_test(r->origin, r->direction, &rec[0], &rec[4], &rec[8], &t, &u, &v)) continue;
                if (!(t >= 0.0f && t <= r->maxDistance)) continue;
                if (any) return 1;
                if (!found || t < best || (t == best && (int32_t)k < best_k)) {
                    found = 1; best = t; best_k = (int32_t)k;
                    out->distance = t; out->primitiveIndex = (int32_t)k;
                    out->coordinates[0] = 1.0f - u - v; out->coordinates[1] = u;
                }
            }
        }
        if (!sp) break;
        node = stack[--sp];
    }
    return found;
}

#ifndef ORC_BRUTE_MAX_TRIS
#define ORC_BRUTE_MAX_TRIS 256u
#endif
void orc_intersect_nearest(orc_scene *s, const orc_ray *rays, size_t n, orc_isect *out, orc_stats *st)
{
    if (s->xn_nodes) { for (size_t i = 0; i < n; ++i) ext_walk_one(s, &rays[i], 0, &out[i], st); return; }
    if (s->ntris <= ORC_BRUTE_MAX_TRIS) { for (size_t i = 0; i < n; ++i) nearest_brute_one(s, &rays[i], &out[i], st); return; }
    const oaccel *A = accel_get(s);
    for (size_t i = 0; i < n; ++i) nearest_bvh_one(s, A, &rays[i], &out[i], st);
}
void orc_intersect_any(orc_scene *s, const orc_ray *rays, size_t n, float *out, orc_stats *st)
{
    if (s->xn_nodes) {   /* (the distance of an any-hit query is never read: -1 = unoccluded, anything else = occluded) */
        for (size_t i = 0; i < n; ++i) out[i] = ext_walk_one(s, &rays[i], 1, NULL, st) ? 0.0f : -1.0f;
        return;
    }
    if (s->ntris <= ORC_BRUTE_MAX_TRIS) { for (size_t i = 0; i < n; ++i) out[i] = any_brute_one(s, &rays[i], st); return; }
    const oaccel *A = accel_get(s);
    for (size_t i = 0; i < n; ++i) out[i] = any_bvh_one(s, A, &rays[i], st);
}

int orc_nearest_f64(const orc_scene *s, const orc_ray *r, double *t_out, double *margin_out)
{
    /* exact-geometry checker in double precision: returns the closest primitive, its t and the
     * "margin" = min over the decision quantities (barycentric slack, t gap to the runner-up),
     * so tests can skip rays whose outcome is legitimately ambiguous in fp32. */
    double o[3] = { r->origin[0], r->origin[1], r->origin[2] };
    double d[3] = { r->direction[0], r->direction[1], r->direction[2] };
    double best = INFINITY, second = INFINITY, best_slack = 0.0; int best_k = -1;
    double min_reject_slack = INFINITY;
    if (r->maxDistance < 0.0f) { *t_out = -1.0; *margin_out = INFINITY; return -1; }
    for (uint32_t k = 0; k < s->ntris; ++k) {
        if ((r->mask & s->material_ids[k]) == 0) continue;
        const float *a = &s->positions[s->indices[k * 3 + 0] * 3];
        const float *b = &s->positions[s->indices[k * 3 + 1] * 3];
        const float *c = &s->positions[s->indices[k * 3 + 2] * 3];
        double e1[3] = { (double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2] };
        double e2[3] = { (double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2] };
        double p[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
        double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
        if (det == 0.0) continue;
        double inv = 1.0 / det;
        double tv[3] = { o[0] - a[0], o[1] - a[1], o[2] - a[2] };
        double u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * inv;
        double q[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
        double v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
        double t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
        double slack = fmin(fmin(u, v), 1.0 - u - v);
        slack = fmin(slack, fmin(t, (double)r->maxDistance - t));
        if (slack >= 0.0) {
            if (t < best) { second = best; best = t; best_k = (int)k; best_slack = slack; }
            else if (t < second) second = t;
        } else if (t > -1e-3) {
            min_reject_slack = fmin(min_reject_slack, -slack);
        }
    }
    *t_out = best_k >= 0 ? best : -1.0;
    double m = min_reject_slack;
    if (best_k >= 0) { m = fmin(m, best_slack); m = fmin(m, second - best); }
    *margin_out = m;
    return best_k;
}

/* =============================== frame loop (a14) =============================== */
int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* One band of rows runs the reference's dispatch order verbatim on band-sized buffers:
 * raygen -> [nearest -> primaryHit -> any -> shadowHit -> swap] x bounces -> accumulate
 * (MetalRenderer.mm:400-515).  Pixels are independent, so banding does not change any value. */
static void render_band(orc_scene *s, const orc_uniforms *u_base, const uint32_t *offsets,
                        uint32_t frame_begin, uint32_t spp, uint32_t bounces,
                        uint32_t y0, uint32_t y1, float *accum, orc_stats *st)
{
    const uint32_t w = u_base->width;
    const size_t base = (size_t)y0 * w;          /* first pixel of the band */
    const size_t n = (size_t)(y1 - y0) * w;
    /* band-local views are indexed with the GLOBAL rayIdx, so shift the base pointers */
    orc_ray *rays = (orc_ray *)malloc(n * sizeof(orc_ray));
    orc_ray *shadow = (orc_ray *)malloc(n * sizeof(orc_ray));
    orc_isect *isects = (orc_isect *)malloc(n * sizeof(orc_isect));
    float *sdist = (float *)malloc(n * sizeof(float));
    float *rt0 = (float *)calloc(n * 4, sizeof(float));
    float *rt1 = (float *)calloc(n * 4, sizeof(float));
    float *acc_tmp = (float *)malloc(n * 4 * sizeof(float));
    for (size_t i = 0; i < n; ++i) { shadow[i].maxDistance = -1.0f; }
    /* views indexed by the GLOBAL rayIdx (element [base] is the band's first pixel) */
#define ORC_SHIFT(T, p, cnt) ((T *)((uintptr_t)(p) - (uintptr_t)(cnt) * sizeof(T)))
    orc_ray *rays_g = ORC_SHIFT(orc_ray, rays, base); orc_ray *shadow_g = ORC_SHIFT(orc_ray, shadow, base);
    orc_isect *isects_g = ORC_SHIFT(orc_isect, isects, base); float *sdist_g = ORC_SHIFT(float, sdist, base);
    for (uint32_t f = frame_begin; f < frame_begin + spp; ++f) {
        orc_uniforms u = *u_base; u.frameIndex = f;
        float *rt[2] = { rt0, rt1 };
        for (uint32_t y = y0; y < y1; ++y)
            for (uint32_t x = 0; x < w; ++x) orc_raygen(&u, offsets, rays_g, ORC_SHIFT(float, rt[0], base * 4), x, y);
        if (st) st->primary_rays += n;
        for (uint32_t b = 0; b < bounces; ++b) {
            if (st && b > 0) for (size_t i = 0; i < n; ++i) st->bounce_rays += rays[i].maxDistance >= 0.0f;
            orc_intersect_nearest(s, rays, n, isects, st);
            for (uint32_t y = y0; y < y1; ++y)
                for (uint32_t x = 0; x < w; ++x)
                    orc_primary_hit(&u, rays_g, shadow_g, isects_g, s, b, offsets, ORC_SHIFT(float, rt[0], base * 4), x, y, st);
            if (st) for (size_t i = 0; i < n; ++i) st->shadow_rays += shadow[i].maxDistance >= 0.0f;
            orc_intersect_any(s, shadow, n, sdist, st);
            for (uint32_t y = y0; y < y1; ++y)
                for (uint32_t x = 0; x < w; ++x)
                    orc_shadow_hit(&u, shadow_g, sdist_g, ORC_SHIFT(float, rt[0], base * 4), ORC_SHIFT(float, rt[1], base * 4), x, y);
            float *t = rt[0]; rt[0] = rt[1]; rt[1] = t;
        }
        for (uint32_t y = y0; y < y1; ++y)
            for (uint32_t x = 0; x < w; ++x) orc_accumulate(&u, ORC_SHIFT(float, rt[0], base * 4), accum, ORC_SHIFT(float, acc_tmp, base * 4), x, y);
        memcpy(accum + base * 4, acc_tmp, n * 4 * sizeof(float));
        /* keep rt0/rt1 roles like the reference's swapped targets */
        rt0 = rt[0]; rt1 = rt[1];
    }
    free(rays); free(shadow); free(isects); free(sdist); free(rt0); free(rt1); free(acc_tmp);
}

int orc_render(orc_scene *s, const orc_uniforms *u_base, const uint32_t *offsets,
               uint32_t frame_begin, uint32_t spp, uint32_t bounces,
               uint32_t row0, uint32_t rows, float *accum, orc_stats *st, int nthreads)
{
    if (!s->xn_nodes && s->ntris > ORC_BRUTE_MAX_TRIS) accel_get(s); /* build outside the parallel region */
    const uint32_t band = 8;
    const uint32_t nb = (rows + band - 1) / band;
    orc_stats total; memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
    {
        orc_stats local; memset(&local, 0, sizeof(local));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (uint32_t b = 0; b < nb; ++b) {
            uint32_t y0 = row0 + b * band, y1 = y0 + band;
            if (y1 > row0 + rows) y1 = row0 + rows;
            render_band(s, u_base, offsets, frame_begin, spp, bounces, y0, y1, accum, st ? &local : NULL);
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.primary_rays += local.primary_rays; total.bounce_rays += local.bounce_rays;
            total.shadow_rays += local.shadow_rays; total.shaded_hits += local.shaded_hits;
            total.tri_tests += local.tri_tests; total.node_visits += local.node_visits;
        }
    }
    if (st) {
        st->primary_rays += total.primary_rays; st->bounce_rays += total.bounce_rays;
        st->shadow_rays += total.shadow_rays; st->shaded_hits += total.shaded_hits;
        st->tri_tests += total.tri_tests; st->node_visits += total.node_visits;
    }
    return 0;
}

/* =============================== post-process (N1) =============================== */
static float srgb(float val)
{
    /* common.h:36-43 */
    if (val < 0.0031308f) val *= 12.92f;
    else val = 1.055f * powf(val, 1.0f / 2.4f) - 0.055f;
    return val;
}
static float aces(float x)
{
    /* common.h:163-171 */
    float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    float v = (x * (a * x + b)) / (x * (c * x + d) + e);
    return fminf(fmaxf(v, 0.0f), 1.0f);
}
void orc_postprocess(const float *accum, uint32_t w, uint32_t h, uint8_t *out, int flip_y)
{
    /* PostProcessing.metal:44-57, then the BGRA8Unorm attachment's round-to-nearest quantisation
     * (MetalRenderer.mm:196); written here as RGBA8.  flip_y=1 puts buffer row h-1 first, the way
     * the reference's quad shows it on screen (PostProcessing.metal:32-41: uv.y=1 at the top). */
    for (uint32_t y = 0; y < h; ++y) {
        uint32_t sy = flip_y ? (h - 1 - y) : y;
        for (uint32_t x = 0; x < w; ++x) {
            const float *p = &accum[((size_t)sy * w + x) * 4];
            uint8_t *o = &out[((size_t)y * w + x) * 4];
            for (int a = 0; a < 3; ++a) {
                float v = srgb(aces(p[a]));
                v = fminf(fmaxf(v, 0.0f), 1.0f);
                o[a] = (uint8_t)(v * 255.0f + 0.5f);
            }
            o[3] = 255;
        }
    }
}
