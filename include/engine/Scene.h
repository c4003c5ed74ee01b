// engine/Scene.h -- toyraygun::Scene (reference src/engine/Scene.h:14-35, Scene.cpp:13-129):
// an unindexed triangle soup in five flat public vectors, filled through addCube / addPlane /
// addAreaLight.  Same names, same layout, same flattening rules (a1 of SURVEY 8a).
#pragma once
#include <bx/math.h>
#include <stdint.h>

#include <vector>

#include "engine/Texture.h"

namespace toyraygun {

class Scene {
public:
    std::vector<bx::Vec3> m_vertexBuffer;      // 3 per triangle, transformed positions
    std::vector<uint32_t> m_indexBuffer;       // 0,1,2,... (identity)
    std::vector<bx::Vec3> m_normalBuffer;      // 3 copies of the transformed, normalised face normal
    std::vector<bx::Vec3> m_colorBuffer;       // 3 copies of the object colour
    std::vector<uint32_t> m_materialIDBuffer;  // 1 per triangle: MATERIAL_DEFAULT / MATERIAL_EMISSIVE

    // beyond the reference ("OBJ and Texture support" is on upstream's to-do list, README.md:18-22): texture coordinates, one
    // (u, v) pair per vertex of the flat buffers above, an albedo texture per triangle (0 = none, k = m_textures[k - 1]) and
    // the textures themselves (borrowed: they must outlive loadScene).  All three stay empty until a textured mesh is added.
    // Project definition of the lookup (the reference has none): the interpolated vertex colour is multiplied by the texel
    // RGB / 255 at x = min(w-1, int(frac(u) w)), y = min(h-1, int(frac(v) h)) -- nearest texel, repeat wrap, row 0 of the
    // image at v = 0; gray textures replicate their value, alpha is ignored.
    std::vector<float> m_uvBuffer;             // 2 per vertex
    std::vector<uint32_t> m_textureIDBuffer;   // 1 per triangle
    std::vector<Texture *> m_textures;

    void addCube(bx::Vec3 color, float *transformMtx);
    void addPlane(bx::Vec3 color, float *transformMtx);
    void addAreaLight(bx::Vec3 color, float *transformMtx);

    // beyond the reference: an arbitrary indexed mesh with per-vertex normals
    void addMesh(const bx::Vec3 *vertices, const bx::Vec3 *normals, const uint32_t *indices, int triangleCount,
                 float *transformMtx, bx::Vec3 color, unsigned int materialID);
    // the same with a colour per vertex (interpolated like the normals, Raytracing.metal:95-112)
    void addMesh(const bx::Vec3 *vertices, const bx::Vec3 *normals, const bx::Vec3 *colors, const uint32_t *indices,
                 int triangleCount, float *transformMtx, unsigned int materialID);

    // the same with texture coordinates and an albedo texture (uvs: 2 floats per vertex)
    void addMesh(const bx::Vec3 *vertices, const bx::Vec3 *normals, const float *uvs, const uint32_t *indices, int triangleCount,
                 float *transformMtx, bx::Vec3 color, unsigned int materialID, Texture *texture);

    // Wavefront OBJ (v / vn / f with v, v/vt, v//vn or v/vt/vn corners, negative indices, polygons as fans).
    // Faces without normals get the flat face normal addGeometry would give them.  Returns the number of
    // triangles added, or -1 if the file cannot be read.  ("To Be Completed" upstream, README.md:18-22.)
    int addObj(const char *path, float *transformMtx, bx::Vec3 color, unsigned int materialID, Texture *texture = nullptr);

protected:
    void padTextureBuffers();   // brings m_uvBuffer / m_textureIDBuffer up to the size of the geometry (untextured triangles)
    uint32_t textureID(Texture *texture);
    void addGeometry(bx::Vec3 *vertices, uint32_t *indices, int triangleCount, float *transformMtx, bx::Vec3 color,
                     unsigned int materialID);
};

}  // namespace toyraygun
