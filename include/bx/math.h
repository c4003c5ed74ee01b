// bx/math.h -- the subset of bkaradzic/bx's math API that ToyRaygun's plugin surface uses
// (reference: lib/bx/include/bx/math.h:14,572-725), so code written against the reference
// (`#include <bx/math.h>`, bx::Vec3, bx::mtxSRT, bx::kPi ...) compiles against this tree unchanged.
// Own implementation; the four functions whose bodies are absent from the reference tree
// (mtxSRT, mtxLookAt, mtxProj, mtxInverse) follow bx's published src/math.cpp (version unpinned,
// SURVEY 8c).  Convention: row-major float[16], row vector times matrix.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace bx {

constexpr float kPi = 3.1415926535897932384626433832795f;
constexpr float kPi2 = 6.2831853071795864769252867665590f;
constexpr float kPiHalf = 1.5707963267948966192313216916398f;

struct Handness { enum Enum { Left, Right }; };

struct Vec3 {
    float x, y, z;
    Vec3() : x(0.0f), y(0.0f), z(0.0f) {}
    Vec3(float _x, float _y, float _z) : x(_x), y(_y), z(_z) {}
    explicit Vec3(float v) : x(v), y(v), z(v) {}
};

float toRad(float deg);
Vec3 add(const Vec3 &a, const Vec3 &b);
Vec3 sub(const Vec3 &a, const Vec3 &b);
Vec3 mul(const Vec3 &a, float s);
float dot(const Vec3 &a, const Vec3 &b);
Vec3 cross(const Vec3 &a, const Vec3 &b);
float length(const Vec3 &a);
Vec3 normalize(const Vec3 &a);
Vec3 calcNormal(const Vec3 &va, const Vec3 &vb, const Vec3 &vc);

void vec4MulMtx(float *result, const float *vec, const float *mat);
void mtxMul(float *result, const float *a, const float *b);
void mtxIdentity(float *result);
void mtxTranspose(float *result, const float *a);
void mtxInverse(float *result, const float *a);
void mtxSRT(float *result, float sx, float sy, float sz, float ax, float ay, float az, float tx, float ty, float tz);
void mtxLookAt(float *result, const Vec3 &eye, const Vec3 &at, const Vec3 &up = Vec3(0.0f, 1.0f, 0.0f),
               Handness::Enum handness = Handness::Left);
void mtxProj(float *result, float fovy, float aspect, float nearPlane, float farPlane, bool homogeneousNdc,
             Handness::Enum handness = Handness::Left);

}  // namespace bx
