// Image.h -- framebuffer of the host surface (reference Image.h:21-96): int RGB triplets, truncating
// setPixel(vec3), ASCII P3 writer; plus a binary P6 writer and a bulk setter for the GPU's RGB8 output.
#pragma once
#include <cstdint>
#include <fstream>
#include <vector>
#include "rt_math.h"

class Image {
    int width, height;
    std::vector<int> data;

public:
    Image(int w, int h) : width(w), height(h), data((size_t)3 * w * h, 0) {}
    int getWidth() const { return width; }
    int getHeight() const { return height; }
    void setPixel(int x, int y, int r, int g, int b) {
        int *p = &data[3 * ((size_t)y * width + x)];
        p[0] = r; p[1] = g; p[2] = b;
    }
    // Image.h:91-95 -- float product truncated towards zero
    void setPixel(int x, int y, glm::vec3 c) { setPixel(x, y, (int)(float)(255 * c.r), (int)(float)(255 * c.g), (int)(float)(255 * c.b)); }
    void setPixel(int x, int y, float r, float g, float b) { setPixel(x, y, glm::vec3(r, g, b)); }
    // bulk load of the renderer's row-major RGB8 output
    void setAll(const uint8_t *rgb8) {
        for (size_t k = 0; k < data.size(); k++) data[k] = rgb8[k];
    }
    int get(int x, int y, int ch) const { return data[3 * ((size_t)y * width + x) + ch]; }
    void writeImage(const char *path) const {  // ASCII P3, same layout as the reference writer (Image.h:40-55)
        std::ofstream f(path);
        f << "P3" << std::endl << width << " " << height << std::endl << 255 << std::endl;
        for (int h = 0; h < height; h++) {
            for (int w = 0; w < width; w++) {
                const int *p = &data[3 * ((size_t)h * width + w)];
                f << p[0] << " " << p[1] << " " << p[2] << "  ";
            }
            f << std::endl;
        }
    }
    void writeImageBinary(const char *path) const {  // P6; 8K frames as P3 are ~400 MB
        std::ofstream f(path, std::ios::binary);
        f << "P6\n" << width << " " << height << "\n255\n";
        std::vector<uint8_t> row((size_t)3 * width);
        for (int h = 0; h < height; h++) {
            for (int k = 0; k < 3 * width; k++) {
                int v = data[(size_t)3 * h * width + k];
                row[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
            f.write((const char *)row.data(), row.size());
        }
    }
};
