// bxmath.cpp -- implementation of include/bx/math.h (own code; algorithms as published in bkaradzic/bx
// src/math.cpp + include/bx/inline/math.inl; see the header for what is pinned and what is not).
#include <bx/math.h>

#include <math.h>
#include <string.h>

namespace bx {

float toRad(float deg) { return deg * kPi / 180.0f; }
Vec3 add(const Vec3 &a, const Vec3 &b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
Vec3 sub(const Vec3 &a, const Vec3 &b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
Vec3 mul(const Vec3 &a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
float dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
Vec3 cross(const Vec3 &a, const Vec3 &b) {
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
float length(const Vec3 &a) { return sqrtf(dot(a, a)); }
Vec3 normalize(const Vec3 &a) {
    const float invLen = 1.0f / length(a);
    return mul(a, invLen);
}
Vec3 calcNormal(const Vec3 &va, const Vec3 &vb, const Vec3 &vc) {
    return normalize(cross(sub(vb, va), sub(vc, va)));
}

void vec4MulMtx(float *r, const float *v, const float *m) {
    const float x = v[0] * m[0] + v[1] * m[4] + v[2] * m[8] + v[3] * m[12];
    const float y = v[0] * m[1] + v[1] * m[5] + v[2] * m[9] + v[3] * m[13];
    const float z = v[0] * m[2] + v[1] * m[6] + v[2] * m[10] + v[3] * m[14];
    const float w = v[0] * m[3] + v[1] * m[7] + v[2] * m[11] + v[3] * m[15];
    r[0] = x; r[1] = y; r[2] = z; r[3] = w;
}
void mtxMul(float *result, const float *a, const float *b) {
    float t[16];
    for (int row = 0; row < 4; ++row) vec4MulMtx(&t[row * 4], &a[row * 4], b);
    memcpy(result, t, sizeof(t));
}
void mtxIdentity(float *m) {
    memset(m, 0, sizeof(float) * 16);
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}
void mtxTranspose(float *result, const float *a) {
    float t[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) t[c * 4 + r] = a[r * 4 + c];
    memcpy(result, t, sizeof(t));
}

// scale, then rotate about X, Y, Z, then translate
void mtxSRT(float *m, float sx, float sy, float sz, float ax, float ay, float az, float tx, float ty, float tz) {
    const float sinx = sinf(ax), cosx = cosf(ax);
    const float siny = sinf(ay), cosy = cosf(ay);
    const float sinz = sinf(az), cosz = cosf(az);
    const float sxsz = sinx * sinz;
    const float cycz = cosy * cosz;
    m[0] = sx * (cycz - sxsz * siny);
    m[1] = sx * -cosx * sinz;
    m[2] = sx * (cosz * siny + cosy * sxsz);
    m[3] = 0.0f;
    m[4] = sy * (cosz * sinx * siny + cosy * sinz);
    m[5] = sy * cosx * cosz;
    m[6] = sy * (siny * sinz - cycz * sinx);
    m[7] = 0.0f;
    m[8] = sz * -cosx * siny;
    m[9] = sz * sinx;
    m[10] = sz * cosx * cosy;
    m[11] = 0.0f;
    m[12] = tx; m[13] = ty; m[14] = tz; m[15] = 1.0f;
}

void mtxLookAt(float *m, const Vec3 &eye, const Vec3 &at, const Vec3 &up, Handness::Enum handness) {
    const Vec3 view = normalize(handness == Handness::Right ? sub(eye, at) : sub(at, eye));
    const Vec3 right = normalize(cross(up, view));
    const Vec3 upv = cross(view, right);
    m[0] = right.x; m[1] = upv.x; m[2] = view.x; m[3] = 0.0f;
    m[4] = right.y; m[5] = upv.y; m[6] = view.y; m[7] = 0.0f;
    m[8] = right.z; m[9] = upv.z; m[10] = view.z; m[11] = 0.0f;
    m[12] = -dot(right, eye);
    m[13] = -dot(upv, eye);
    m[14] = -dot(view, eye);
    m[15] = 1.0f;
}

void mtxProj(float *m, float fovy, float aspect, float nearPlane, float farPlane, bool homogeneousNdc,
             Handness::Enum handness) {
    const float height = 1.0f / tanf(toRad(fovy) * 0.5f);
    const float width = height * 1.0f / aspect;
    const float diff = farPlane - nearPlane;
    const float aa = homogeneousNdc ? (farPlane + nearPlane) / diff : farPlane / diff;
    const float bb = homogeneousNdc ? (2.0f * farPlane * nearPlane) / diff : nearPlane * aa;
    const bool rh = handness == Handness::Right;
    memset(m, 0, sizeof(float) * 16);
    m[0] = width;
    m[5] = height;
    m[8] = 0.0f;
    m[9] = 0.0f;
    m[10] = rh ? -aa : aa;
    m[11] = rh ? -1.0f : 1.0f;
    m[14] = -bb;
}

// cofactor expansion
void mtxInverse(float *result, const float *a) {
    const float xx = a[0], xy = a[1], xz = a[2], xw = a[3];
    const float yx = a[4], yy = a[5], yz = a[6], yw = a[7];
    const float zx = a[8], zy = a[9], zz = a[10], zw = a[11];
    const float wx = a[12], wy = a[13], wz = a[14], ww = a[15];
    float det = 0.0f;
    det += xx * (yy * (zz * ww - zw * wz) - yz * (zy * ww - zw * wy) + yw * (zy * wz - zz * wy));
    det -= xy * (yx * (zz * ww - zw * wz) - yz * (zx * ww - zw * wx) + yw * (zx * wz - zz * wx));
    det += xz * (yx * (zy * ww - zw * wy) - yy * (zx * ww - zw * wx) + yw * (zx * wy - zy * wx));
    det -= xw * (yx * (zy * wz - zz * wy) - yy * (zx * wz - zz * wx) + yz * (zx * wy - zy * wx));
    const float invDet = 1.0f / det;
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
    memcpy(result, t, sizeof(t));
}

}  // namespace bx
