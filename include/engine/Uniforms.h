// engine/Uniforms.h -- host mirror of the shader Uniforms block (reference src/engine/Uniforms.h:19-41,
// src/engine/Metal/MetalUniforms.h:16-60).  UniformFloat3 is 16 bytes, UniformFloat4x4 stores columns
// and transposes on set/get exactly like the Metal variant, so sizeof(Uniforms) == 176 and the struct
// can be handed to trg_set_uniforms() as is.
#pragma once
#include <bx/math.h>

#include "engine/Engine.h"

namespace toyraygun {

struct UniformFloat3 {
    float data[4];
    bx::Vec3 get() { return bx::Vec3(data[0], data[1], data[2]); }
    void set(bx::Vec3 v) { data[0] = v.x; data[1] = v.y; data[2] = v.z; }
};

struct UniformFloat4x4 {
    float columns[4][4];
    void get(float *mtxOut) {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) mtxOut[r * 4 + c] = columns[c][r];  // transpose, MetalUniforms.h:33-44
    }
    void set(float *mtxIn) {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) columns[c][r] = mtxIn[r * 4 + c];  // transpose, MetalUniforms.h:49-59
    }
};

struct Camera {
    UniformFloat3 position;
    UniformFloat4x4 invViewProjMtx;
};

struct AreaLight {
    UniformFloat3 position, forward, right, up, color;
};

struct Uniforms {
    unsigned int width, height, frameIndex, _pad;
    Camera camera;
    AreaLight light;
};

static_assert(sizeof(Uniforms) == 176, "Uniforms must match the 176-byte shader block");

}  // namespace toyraygun
