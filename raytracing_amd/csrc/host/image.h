// Float RGB images for the host front end: BMP reader/writer and a scan-line
// PIZ OpenEXR reader (half RGBA), plus the bump-map differentiation.
// Semantics follow src/bitmap.cpp (BMP :101-220, EXR :222-252, Differentiate :312-326).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace qdh {

struct Image {
    uint32_t w = 0, h = 0;
    std::vector<float> rgb;            // 3 floats per texel, row-major x + y*w
    bool ok() const { return !rgb.empty(); }
    // Bitmap::GetPixel: black outside the image (src/bitmap.cpp:44-50)
    void get(uint32_t x, uint32_t y, float out[3]) const {
        if (rgb.empty() || x >= w || y >= h) { out[0] = out[1] = out[2] = 0.f; return; }
        const float* p = &rgb[(size_t(x) + size_t(y) * w) * 3];
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
    }
    void set(uint32_t x, uint32_t y, float r, float g, float b) {
        if (rgb.empty() || x >= w || y >= h) return;
        float* p = &rgb[(size_t(x) + size_t(y) * w) * 3];
        p[0] = r; p[1] = g; p[2] = b;
    }
    void make(uint32_t w_, uint32_t h_) { w = w_; h = h_; rgb.assign(size_t(w_) * h_ * 3, 0.f); }
};

bool load_bmp(const char* path, Image& img, std::string& err);
bool load_exr(const char* path, Image& img, std::string& err);
bool load_image(const char* path, Image& img, std::string& err);   // by extension, like Bitmap::LoadImage
bool save_bmp(const char* path, const float* rgb, int w, int h, size_t pitch_floats, std::string& err);
bool save_exr(const char* path, const float* rgb, int w, int h, size_t pitch_floats, std::string& err);
uint16_t float_to_half(float f);
void differentiate(Image& img);                                     // Bitmap::Differentiate
extern int g_differentiate_hook_calls;
extern int (*g_differentiate_hook)(uint32_t, uint32_t, float*);     // device version of it (qdh_set_prep_hooks), or null
unsigned convert_to_8bit(double x);                                 // ConvertTo8Bit, src/utils.h:22,34
bool file_exists(const std::string& path);

}  // namespace qdh
