// decodes a JPEG with sfm_jpeg.hpp (through the drivers' imread) and writes it as binary PPM (RGB) / PGM: tests/test_jpeg_cpu.py
#include "sfm_features.hpp"
#include <cstdio>
int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.jpg out.ppm\n", argv[0]); return 2; }
    const sfm::Image img = sfm::imread(argv[1]);
    if (img.empty()) { fprintf(stderr, "decode failed\n"); return 1; }
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    fprintf(f, "%s\n%d %d\n255\n", img.channels == 3 ? "P6" : "P5", img.cols, img.rows);
    std::vector<uint8_t> row((size_t)img.cols * img.channels);
    for (int y = 0; y < img.rows; ++y) {
        for (int x = 0; x < img.cols; ++x)
            if (img.channels == 3) { const uint8_t* p = img.at(y, x); row[3 * x] = p[2]; row[3 * x + 1] = p[1]; row[3 * x + 2] = p[0]; }
            else row[x] = *img.at(y, x);
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return 0;
}
