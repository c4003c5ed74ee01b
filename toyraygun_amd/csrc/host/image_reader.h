// image_reader.h -- PNG / PPM decoder behind toyraygun::Texture::loadFile (see image_reader.cpp).
#pragma once
#include <stdint.h>

namespace trg_host {
// malloc'ed bytes (free() them), rows top to bottom, `channels` bytes per pixel as stored in the file; nullptr on failure
uint8_t *read_image(const char *path, int *width, int *height, int *channels);
}
