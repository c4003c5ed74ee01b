// parser_fuzz.cpp -- mutation fuzz of the two file parsers behind the plugin surface (Texture::loadFile -> trg_host::read_image: PNG / PPM;
// Scene::addObj: Wavefront OBJ), built with -fsanitize=address,undefined by tests/test_host_surface.py (CPU only).  Seeds are written by the
// test; every iteration flips / overwrites / truncates a few bytes of a seed and parses the result: any answer is fine, a sanitizer report is not.
//   parser_fuzz <iterations> <seed file> ...        (.obj seeds go to Scene::addObj, everything else to read_image)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "engine/Scene.h"
#include "image_reader.h"

static uint32_t rng_state = 0x5EED0005u;
static uint32_t rnd() { rng_state = rng_state * 747796405u + 2891336453u; uint32_t w = ((rng_state >> ((rng_state >> 28) + 4u)) ^ rng_state) * 277803737u; return (w >> 22) ^ w; }

static std::vector<uint8_t> slurp(const char *path) {
    std::vector<uint8_t> b;
    if (FILE *f = fopen(path, "rb")) { uint8_t buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n); fclose(f); }
    return b;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int iters = atoi(argv[1]);
    const std::string tmp = std::string(argv[2]) + ".mut";
    long parsed = 0, accepted = 0;
    for (int s = 2; s < argc; ++s) {
        const std::vector<uint8_t> seed = slurp(argv[s]);
        const bool obj = strlen(argv[s]) > 4 && !strcmp(argv[s] + strlen(argv[s]) - 4, ".obj");
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> m = seed;
            const int edits = 1 + (int)(rnd() % 6u);
            for (int e = 0; e < edits && !m.empty(); ++e) {
                const size_t at = rnd() % m.size();
                switch (rnd() % 5u) {
                    case 0: m[at] ^= (uint8_t)(1u << (rnd() % 8u)); break;
                    case 1: m[at] = (uint8_t)rnd(); break;
                    case 2: m.resize(at); break;                                                     // truncate
                    case 3: { const uint32_t v = (rnd() & 1u) ? 0xFFFFFFFFu : rnd(); for (size_t k = 0; k < 4 && at + k < m.size(); ++k) m[at + k] = (uint8_t)(v >> (24 - 8 * k)); break; }   // a big-endian length / size field
                    default: { const size_t from = rnd() % m.size(), len = rnd() % 24u; const std::vector<uint8_t> piece(m.begin() + (long)from, m.begin() + (long)(from + len < m.size() ? from + len : m.size())); m.insert(m.begin() + (long)at, piece.begin(), piece.end()); break; }   // splice a short piece of the file in
                }
            }
            if (FILE *f = fopen(tmp.c_str(), "wb")) { if (!m.empty()) fwrite(m.data(), 1, m.size(), f); fclose(f); }
            ++parsed;
            if (obj) {
                toyraygun::Scene sc;
                float mtx[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
                if (sc.addObj(tmp.c_str(), mtx, bx::Vec3(1.0f, 1.0f, 1.0f), 1u) > 0) ++accepted;
            } else {
                int w = 0, h = 0, c = 0;
                if (uint8_t *px = trg_host::read_image(tmp.c_str(), &w, &h, &c)) {
                    volatile uint8_t sink = px[0] ^ px[(size_t)w * h * c - 1];   // the buffer really has w * h * c bytes
                    (void)sink;
                    free(px);
                    ++accepted;
                }
            }
        }
    }
    remove(tmp.c_str());
    printf("parser_fuzz: %ld inputs parsed, %ld accepted\n", parsed, accepted);
    return 0;
}
