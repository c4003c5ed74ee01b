// Scene.cpp -- toyraygun::Scene geometry builders (a1 of SURVEY 8a; reference src/engine/Scene.cpp:13-129).
#include "engine/Scene.h"

#include "engine/Renderer.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace toyraygun {
namespace {

// unit cube corners, bit 0 = +x, bit 1 = +y, bit 2 = +z (Scene.cpp:13-22)
bx::Vec3 corner(int i) { return bx::Vec3((i & 1) ? 0.5f : -0.5f, (i & 2) ? 0.5f : -0.5f, (i & 4) ? 0.5f : -0.5f); }

bx::Vec3 transformed(const bx::Vec3 &v, const float *mtx, float w) {
    const float in[4] = { v.x, v.y, v.z, w };
    float out[4];
    bx::vec4MulMtx(out, in, mtx);
    return bx::Vec3(out[0], out[1], out[2]);
}

// the quad both addPlane and addAreaLight use: the cube's bottom face (corners 0,1,5,4), two
// triangles wound 0-2-1 / 0-3-2 (Scene.cpp:60-92)
void quad(bx::Vec3 *verts, uint32_t *tris) {
    const int sel[4] = { 0, 1, 5, 4 };
    for (int i = 0; i < 4; ++i) verts[i] = corner(sel[i]);
    const uint32_t t[6] = { 0, 2, 1, 0, 3, 2 };
    for (int i = 0; i < 6; ++i) tris[i] = t[i];
}

}  // namespace

void Scene::addCube(bx::Vec3 color, float *transformMtx) {
    bx::Vec3 verts[8];
    for (int i = 0; i < 8; ++i) verts[i] = corner(i);
    // face order and winding of Scene.cpp:37-55: -x, +x, -y, +y, -z, +z
    uint32_t tris[36] = { 0, 4, 6, 0, 6, 2, 1, 3, 7, 1, 7, 5, 0, 1, 5, 0, 5, 4,
                          2, 6, 7, 2, 7, 3, 0, 2, 3, 0, 3, 1, 4, 5, 7, 4, 7, 6 };
    addGeometry(verts, tris, 12, transformMtx, color, MATERIAL_DEFAULT);
}

void Scene::addPlane(bx::Vec3 color, float *transformMtx) {
    bx::Vec3 verts[4];
    uint32_t tris[6];
    quad(verts, tris);
    addGeometry(verts, tris, 2, transformMtx, color, MATERIAL_DEFAULT);
}

void Scene::addAreaLight(bx::Vec3 color, float *transformMtx) {
    bx::Vec3 verts[4];
    uint32_t tris[6];
    quad(verts, tris);
    addGeometry(verts, tris, 2, transformMtx, color, MATERIAL_EMISSIVE);
}

// Flattening rule (Scene.cpp:102-129): per triangle, the face normal is taken from the UNtransformed
// corners, pushed through the matrix with w = 0 and re-normalised; each corner gets its own vertex.
void Scene::addGeometry(bx::Vec3 *vertices, uint32_t *indices, int triangleCount, float *transformMtx, bx::Vec3 color,
                        unsigned int materialID) {
    for (int t = 0; t < triangleCount; ++t) {
        const uint32_t *tri = &indices[t * 3];
        const bx::Vec3 faceNormal = bx::calcNormal(vertices[tri[0]], vertices[tri[1]], vertices[tri[2]]);
        for (int c = 0; c < 3; ++c) {
            m_vertexBuffer.push_back(transformed(vertices[tri[c]], transformMtx, 1.0f));
            m_indexBuffer.push_back((uint32_t)(m_vertexBuffer.size() - 1));
            m_normalBuffer.push_back(bx::normalize(transformed(faceNormal, transformMtx, 0.0f)));
            m_colorBuffer.push_back(color);
        }
        m_materialIDBuffer.push_back(materialID);
    }
}

void Scene::addMesh(const bx::Vec3 *vertices, const bx::Vec3 *normals, const uint32_t *indices, int triangleCount,
                    float *transformMtx, bx::Vec3 color, unsigned int materialID) {
    for (int t = 0; t < triangleCount; ++t) {
        const uint32_t *tri = &indices[t * 3];
        for (int c = 0; c < 3; ++c) {
            m_vertexBuffer.push_back(transformed(vertices[tri[c]], transformMtx, 1.0f));
            m_indexBuffer.push_back((uint32_t)(m_vertexBuffer.size() - 1));
            m_normalBuffer.push_back(bx::normalize(transformed(normals[tri[c]], transformMtx, 0.0f)));
            m_colorBuffer.push_back(color);
        }
        m_materialIDBuffer.push_back(materialID);
    }
}

void Scene::addMesh(const bx::Vec3 *vertices, const bx::Vec3 *normals, const bx::Vec3 *colors, const uint32_t *indices,
                    int triangleCount, float *transformMtx, unsigned int materialID) {
    for (int t = 0; t < triangleCount; ++t) {
        const uint32_t *tri = &indices[t * 3];
        for (int c = 0; c < 3; ++c) {
            m_vertexBuffer.push_back(transformed(vertices[tri[c]], transformMtx, 1.0f));
            m_indexBuffer.push_back((uint32_t)(m_vertexBuffer.size() - 1));
            m_normalBuffer.push_back(bx::normalize(transformed(normals[tri[c]], transformMtx, 0.0f)));
            m_colorBuffer.push_back(colors[tri[c]]);
        }
        m_materialIDBuffer.push_back(materialID);
    }
}

void Scene::padTextureBuffers() {
    m_uvBuffer.resize(m_vertexBuffer.size() * 2, 0.0f);
    m_textureIDBuffer.resize(m_materialIDBuffer.size(), 0u);
}
uint32_t Scene::textureID(Texture *texture) {
    if (!texture) return 0u;
    for (size_t k = 0; k < m_textures.size(); ++k)
        if (m_textures[k] == texture) return (uint32_t)k + 1u;
    m_textures.push_back(texture);
    return (uint32_t)m_textures.size();
}

void Scene::addMesh(const bx::Vec3 *vertices, const bx::Vec3 *normals, const float *uvs, const uint32_t *indices, int triangleCount,
                    float *transformMtx, bx::Vec3 color, unsigned int materialID, Texture *texture) {
    padTextureBuffers();   // whatever was added before has no texture
    const uint32_t id = textureID(texture);
    addMesh(vertices, normals, indices, triangleCount, transformMtx, color, materialID);
    for (int t = 0; t < triangleCount; ++t) {
        for (int c = 0; c < 3; ++c) {
            const uint32_t v = indices[t * 3 + c];
            m_uvBuffer.push_back(uvs ? uvs[v * 2] : 0.0f);
            m_uvBuffer.push_back(uvs ? uvs[v * 2 + 1] : 0.0f);
        }
        m_textureIDBuffer.push_back(id);
    }
}

int Scene::addObj(const char *path, float *transformMtx, bx::Vec3 color, unsigned int materialID, Texture *texture) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    std::vector<bx::Vec3> pos, nrm;
    std::vector<float> tex;   // vt u v
    int added = 0;
    char line[1024];
    while (fgets(line, sizeof(line), f)) {
        const char *p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            float x = 0, y = 0, z = 0;
            if (sscanf(p + 1, "%f %f %f", &x, &y, &z) == 3) pos.push_back(bx::Vec3(x, y, z));
        } else if (p[0] == 'v' && p[1] == 't') {
            float u = 0, v = 0;
            if (sscanf(p + 2, "%f %f", &u, &v) >= 1) { tex.push_back(u); tex.push_back(v); }
        } else if (p[0] == 'v' && p[1] == 'n') {
            float x = 0, y = 0, z = 0;
            if (sscanf(p + 2, "%f %f %f", &x, &y, &z) == 3) nrm.push_back(bx::Vec3(x, y, z));
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            // corners: v, v/vt, v//vn, v/vt/vn
            int vi[64], ni[64], ti[64], n = 0;
            char *q = const_cast<char *>(p + 1);
            while (n < 64) {
                while (*q == ' ' || *q == '\t') ++q;
                if (*q == 0 || *q == '\n' || *q == '\r' || *q == '#') break;
                char *end;
                long v = strtol(q, &end, 10), t = 0, nn = 0;
                if (end == q) break;
                q = end;
                if (*q == '/') {
                    ++q;
                    if (*q != '/') { t = strtol(q, &end, 10); q = end; }
                    if (*q == '/') { ++q; nn = strtol(q, &end, 10); q = end; }
                }
                ti[n] = t == 0 ? -1 : (int)(t < 0 ? (long)(tex.size() / 2) + t : t - 1);
                vi[n] = (int)(v < 0 ? (long)pos.size() + v : v - 1);
                ni[n] = nn == 0 ? -1 : (int)(nn < 0 ? (long)nrm.size() + nn : nn - 1);
                ++n;
            }
            for (int k = 1; k + 1 < n; ++k) {  // triangle fan
                const int c3[3] = { 0, k, k + 1 };
                bool ok = true, have_n = true;
                for (int j = 0; j < 3; ++j) {
                    if (vi[c3[j]] < 0 || vi[c3[j]] >= (int)pos.size()) ok = false;
                    if (ni[c3[j]] < 0 || ni[c3[j]] >= (int)nrm.size()) have_n = false;
                }
                if (!ok) continue;
                bx::Vec3 v3[3] = { pos[vi[c3[0]]], pos[vi[c3[1]]], pos[vi[c3[2]]] };
                uint32_t idx[3] = { 0, 1, 2 };
                bool have_t = texture != nullptr;
                for (int j = 0; j < 3; ++j)
                    if (ti[c3[j]] < 0 || ti[c3[j]] >= (int)(tex.size() / 2)) have_t = false;
                if (have_t) padTextureBuffers();
                if (have_n) {
                    bx::Vec3 n3[3] = { nrm[ni[c3[0]]], nrm[ni[c3[1]]], nrm[ni[c3[2]]] };
                    addMesh(v3, n3, idx, 1, transformMtx, color, materialID);
                } else {
                    addGeometry(v3, idx, 1, transformMtx, color, materialID);
                }
                if (have_t) {   // a face with texture coordinates of a textured OBJ
                    for (int j = 0; j < 3; ++j) { m_uvBuffer.push_back(tex[ti[c3[j]] * 2]); m_uvBuffer.push_back(tex[ti[c3[j]] * 2 + 1]); }
                    m_textureIDBuffer.push_back(textureID(texture));
                }
                ++added;
            }
        }
    }
    fclose(f);
    return added;
}

}  // namespace toyraygun
