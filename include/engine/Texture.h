// engine/Texture.h -- toyraygun::Texture (reference src/engine/Texture.h:14-36).  Only
// generateRandomTexture is on the hot path (the per-pixel Halton offsets, Texture.cpp:16-29); the
// reference fills it from unseeded libc rand(), here the bytes come from a seeded hash so renders
// are reproducible.  loadFile() (Texture.cpp:39-48, stb_image in the reference) decodes 8-bit PNG and binary
// PPM / PGM with the in-tree reader (csrc/host/image_reader.cpp); a loaded texture can be given to Scene::addMesh as the
// albedo map of a mesh with texture coordinates ("OBJ and Texture support", README.md:18-22, upstream's to-do list).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace toyraygun {

class Texture {
public:
    static Texture generateRandomTexture(int width, int height, int channels);
    static Texture generateRandomTexture(int width, int height, int channels, uint32_t seed);

    virtual ~Texture() {}
    virtual void init(int width, int height, int channels);
    virtual bool loadFile(std::string path);  // 8-bit PNG (gray, gray+alpha, RGB, RGBA; non-interlaced) or binary PPM / PGM
    virtual void destroy();
    virtual uint8_t *getBufferPointer();
    virtual size_t getBufferSize();
    virtual size_t getBufferStride();
    virtual int getWidth();
    virtual int getHeight();
    virtual int getChannels();

protected:
    void *m_data = nullptr;
    int m_width = 0, m_height = 0, m_channels = 0;
};

}  // namespace toyraygun
