// Texture.cpp -- toyraygun::Texture (include/engine/Texture.h).
#include "engine/Texture.h"

#include <stdlib.h>
#include <string.h>

#include "image_reader.h"

namespace toyraygun {
namespace {
uint32_t pcg_hash32(uint32_t v) {
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
}  // namespace

Texture Texture::generateRandomTexture(int width, int height, int channels) {
    return generateRandomTexture(width, height, channels, 0x5EED0001u);
}

// Reference: every BYTE = (uint8)(rand() % 2^20) (Texture.cpp:23-26), read back as one little-endian
// u32 per pixel when channels == 4.  Here: u32 pixel p = pcg_hash32(seed ^ p), stored little-endian;
// other channel counts hash the byte index.
Texture Texture::generateRandomTexture(int width, int height, int channels, uint32_t seed) {
    Texture t;
    t.init(width, height, channels);
    uint8_t *bytes = t.getBufferPointer();
    const size_t n = (size_t)width * height;
    if (channels == 4) {
        for (size_t p = 0; p < n; ++p) {
            const uint32_t v = pcg_hash32(seed ^ (uint32_t)p);
            bytes[p * 4 + 0] = (uint8_t)(v);
            bytes[p * 4 + 1] = (uint8_t)(v >> 8);
            bytes[p * 4 + 2] = (uint8_t)(v >> 16);
            bytes[p * 4 + 3] = (uint8_t)(v >> 24);
        }
    } else {
        for (size_t i = 0; i < n * (size_t)channels; ++i) bytes[i] = (uint8_t)pcg_hash32(seed ^ (uint32_t)i);
    }
    return t;
}

void Texture::init(int width, int height, int channels) {
    m_data = malloc((size_t)width * height * channels);
    m_width = width; m_height = height; m_channels = channels;
}
bool Texture::loadFile(std::string path) {
    destroy();
    m_data = trg_host::read_image(path.c_str(), &m_width, &m_height, &m_channels);   // like stbi_load(path, &w, &h, &c, 0)
    return m_data != nullptr;
}
void Texture::destroy() {
    if (m_data) free(m_data);
    m_data = nullptr;
}
uint8_t *Texture::getBufferPointer() { return (uint8_t *)m_data; }
size_t Texture::getBufferSize() { return (size_t)m_width * m_height * m_channels; }
size_t Texture::getBufferStride() { return (size_t)m_width * m_channels; }
int Texture::getWidth() { return m_width; }
int Texture::getHeight() { return m_height; }
int Texture::getChannels() { return m_channels; }

}  // namespace toyraygun
