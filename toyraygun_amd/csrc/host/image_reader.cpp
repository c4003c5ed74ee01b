// image_reader.cpp -- the decoder behind toyraygun::Texture::loadFile (include/engine/Texture.h).
// The reference calls stb_image here (src/engine/Texture.cpp:39-48), a dependency this tree does not carry; this is a small
// decoder of its own for the two formats the project uses: PNG (8-bit gray / gray+alpha / RGB / RGBA, non-interlaced; zlib
// inflate with stored, fixed and dynamic Huffman blocks; all five scanline filters) and binary PPM / PGM (P6 / P5, maxval 255).
// Output like stbi_load(..., 0): the file's own channel count, rows top to bottom, tightly packed bytes.
#include "image_reader.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace trg_host {
namespace {

struct BitReader {
    const uint8_t *p, *end;
    uint32_t buf = 0;
    int n = 0;
    bool bad = false;
    uint32_t bits(int k) {
        while (n < k) {
            if (p >= end) { bad = true; return 0; }
            buf |= (uint32_t)(*p++) << n;
            n += 8;
        }
        const uint32_t v = buf & ((k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1u));
        buf = k == 32 ? 0 : buf >> k;
        n -= k;
        return v;
    }
    void align() { buf = 0; n = 0; }
};

struct Huffman {
    uint16_t count[16], symbol[288];
    void build(const uint8_t *lengths, int n) {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = (uint16_t)(offs[i] + count[i]);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= (int)br.bits(1);
            if (br.bad) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

// `cap`: the decoder fails as soon as the output would exceed it (a few bytes of deflate can expand a thousandfold)
bool inflate_raw(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t cap) {
    static const uint16_t lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
    static const uint16_t lext[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
    static const uint16_t dbase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
    static const uint16_t dext[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
    BitReader br{ src, src + n };
    for (;;) {
        const uint32_t last = br.bits(1), type = br.bits(2);
        if (br.bad) return false;
        if (type == 0) {
            br.align();
            if (br.end - br.p < 4) return false;
            const uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((len ^ 0xFFFFu) != nlen || (size_t)(br.end - br.p) < len || out.size() + len > cap) return false;
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lengths[i] = 8;
                for (; i < 256; ++i) lengths[i] = 9;
                for (; i < 280; ++i) lengths[i] = 7;
                for (; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (br.bad || nlen > 286 || ndist > 30) return false;
                static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                uint8_t cl[19];
                memset(cl, 0, sizeof(cl));
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman lencode;
                lencode.build(cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = lencode.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                    int rep, val = 0;
                    if (sym == 16) { if (idx == 0) return false; val = lengths[idx - 1]; rep = 3 + (int)br.bits(2); }
                    else if (sym == 17) rep = 3 + (int)br.bits(3);
                    else rep = 11 + (int)br.bits(7);
                    if (br.bad || idx + rep > nlen + ndist) return false;
                    while (rep--) lengths[idx++] = (uint8_t)val;
                }
                lit.build(lengths, nlen);
                dist.build(lengths + nlen, ndist);
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) return false;
                if (sym < 256) { if (out.size() >= cap) return false; out.push_back((uint8_t)sym); continue; }
                if (sym == 256) break;
                const int li = sym - 257;
                if (li >= 29) return false;
                const int len = lbase[li] + (int)br.bits(lext[li]);
                const int ds = dist.decode(br);
                if (ds < 0 || ds >= 30) return false;
                const size_t d = dbase[ds] + br.bits(dext[ds]);
                if (br.bad || d > out.size() || out.size() + (size_t)len > cap) return false;
                const size_t start = out.size() - d;
                for (int k = 0; k < len; ++k) out.push_back(out[start + k]);
            }
        } else {
            return false;
        }
        if (last) return true;
    }
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

uint8_t *decode_png(const std::vector<uint8_t> &file, int *w, int *h, int *channels) {
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) return nullptr;
    size_t pos = 8;
    uint32_t width = 0, height = 0;
    int ch = 0;
    std::vector<uint8_t> z;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
        if (pos + 12 + (size_t)len > file.size()) return nullptr;
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) return nullptr;
            width = be32(data); height = be32(data + 4);
            const int depth = data[8], ctype = data[9], interlace = data[12];
            if (depth != 8 || interlace != 0) return nullptr;
            ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;   // no palette
            if (!ch || width == 0 || height == 0 || width > 32768 || height > 32768) return nullptr;
        } else if (!memcmp(type, "IDAT", 4)) {
            z.insert(z.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!ch || z.size() < 6) return nullptr;
    // The header alone must not size anything: a 60-byte file can claim 32768 x 32768 x 4.  Deflate expands at most ~1032-fold, so the
    // compressed size bounds what is reserved; the decoder stops at the size the header implies; and a picture over 1 GiB decoded is
    // refused outright (this library is built without exceptions: a failed allocation would end the process, not the call).
    const size_t stride = (size_t)width * ch, expected = (size_t)height * (stride + 1);
    if (expected > ((size_t)1 << 30)) return nullptr;
    std::vector<uint8_t> raw;
    raw.reserve(std::min(expected, z.size() * 1032u + 64u));
    if (!inflate_raw(z.data() + 2, z.size() - 2, raw, expected)) return nullptr;   // 2-byte zlib header; the adler32 trailer is not checked
    if (raw.size() < expected) return nullptr;
    uint8_t *out = (uint8_t *)malloc((size_t)height * stride);
    if (!out) return nullptr;
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t *line = &raw[(size_t)y * (stride + 1)];
        const int filter = line[0];
        uint8_t *cur = out + (size_t)y * stride;
        const uint8_t *prev = y ? cur - stride : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev ? prev[i] : 0, c = (prev && i >= (size_t)ch) ? prev[i - ch] : 0;
            int v = line[1 + i];
            switch (filter) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: free(out); return nullptr;
            }
            cur[i] = (uint8_t)v;
        }
    }
    *w = (int)width; *h = (int)height; *channels = ch;
    return out;
}

uint8_t *decode_pnm(const std::vector<uint8_t> &file, int *w, int *h, int *channels) {
    if (file.size() < 7 || file[0] != 'P' || (file[1] != '6' && file[1] != '5')) return nullptr;
    const int ch = file[1] == '6' ? 3 : 1;
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < file.size()) {
        const uint8_t c = file[pos];
        if (c == '#') { while (pos < file.size() && file[pos] != '\n') ++pos; continue; }
        if (c == ' ' || c == '\n' || c == '\r' || c == '\t') { ++pos; continue; }
        if (c < '0' || c > '9') return nullptr;
        int v = 0;
        while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') { v = v * 10 + (file[pos] - '0'); ++pos; if (v > 1 << 20) return nullptr; }
        vals[got++] = v;
    }
    if (got != 3 || vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0) return nullptr;
    ++pos;   // the single whitespace byte after maxval
    const size_t bytes = (size_t)vals[0] * vals[1] * ch;
    if (pos + bytes > file.size()) return nullptr;
    uint8_t *out = (uint8_t *)malloc(bytes);
    if (!out) return nullptr;
    memcpy(out, &file[pos], bytes);
    *w = vals[0]; *h = vals[1]; *channels = ch;
    return out;
}

}  // namespace

uint8_t *read_image(const char *path, int *width, int *height, int *channels) {
    FILE *f = fopen(path, "rb");
    if (!f) return nullptr;
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    uint8_t *out = decode_png(file, width, height, channels);
    if (!out) out = decode_pnm(file, width, height, channels);
    return out;
}

}  // namespace trg_host
