// BmpReader.h -- minimal uncompressed-BMP decoder for the four PBR maps of the sample scene.
// Produces what the reference's texture fetch expects (bmpmini.hpp:65-88 as used by Textures.h:156-166):
// rows top-down, bytes in file (BGR) order, no row padding.  Only the fields the reference reads are
// interpreted: data offset, width, height, bits per pixel, compression, palette size.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

struct BmpImage {
    int width = 0, height = 0, channels = 0;
    std::vector<uint8_t> pixels;  // top-down, BGR
};

inline BmpImage readBmp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("readBmp: cannot open " + path);
    uint8_t h[54];
    f.read((char *)h, 54);
    if (f.gcount() != 54 || h[0] != 'B' || h[1] != 'M') throw std::runtime_error("readBmp: not a BMP: " + path);
    auto u32 = [&](int o) { uint32_t v; std::memcpy(&v, h + o, 4); return v; };
    auto u16 = [&](int o) { uint16_t v; std::memcpy(&v, h + o, 2); return v; };
    const uint32_t offset = u32(10);
    const int32_t w = (int32_t)u32(18), ht = (int32_t)u32(22);
    const int bpp = u16(28);
    if (u32(30) != 0 || u32(46) != 0) throw std::runtime_error("readBmp: compressed / palettised BMP unsupported: " + path);
    if (w <= 0 || ht <= 0 || (bpp != 24 && bpp != 32 && bpp != 8)) throw std::runtime_error("readBmp: unsupported geometry: " + path);
    const size_t rowBytes = (size_t)w * bpp / 8, padded = (((size_t)w * bpp + 31) / 32) * 4;
    std::vector<uint8_t> raw(padded * ht);
    f.seekg(offset, std::ios::beg);
    f.read((char *)raw.data(), raw.size());
    if ((size_t)f.gcount() != raw.size()) throw std::runtime_error("readBmp: truncated file: " + path);
    BmpImage img;
    img.width = w; img.height = ht; img.channels = bpp / 8;
    img.pixels.resize(rowBytes * ht);
    for (int r = 0; r < ht; r++)  // file is bottom-up: output row r is file row (ht-1-r)
        std::memcpy(&img.pixels[(size_t)r * rowBytes], &raw[(size_t)(ht - 1 - r) * padded], rowBytes);
    return img;
}
