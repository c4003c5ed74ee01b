// png_writer.h -- minimal PNG encoder (8-bit RGBA, stored/uncompressed deflate blocks); the headless
// replacement for the reference's swapchain present (MetalRenderer.mm:523-547).
#pragma once
#include <stdint.h>

namespace trg_host {
bool write_png_rgba8(const char *path, const uint8_t *rgba, int width, int height);
}
