// BMP reader/writer, scan-line OpenEXR reader (uncompressed + PIZ, HALF channels)
// and bump-map differentiation for the host front end.
//
// Behaviour follows src/bitmap.cpp of the reference:
//   LoadBMP  :101-179  (8/24/32 bpp, rows bottom-up, BGR, 8-bit via palette built by Color(unsigned))
//   SaveBMP  :181-220  (24 bpp, bottom-up, ConvertTo8Bit quantisation, never sRGB)
//   LoadEXR  :222-252  (the reference calls OpenEXR 2.0.1 RgbaInputFile; OpenEXR is not
//                       available here, so the PIZ codec below is written from the published
//                       format description: bitmap -> LUT, canonical Huffman, 2-D wavelet)
//   Differentiate :312-326
#include "image.h"

#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace qdh {

int (*g_differentiate_hook)(uint32_t, uint32_t, float*) = nullptr;   // qdh_set_prep_hooks
int g_differentiate_hook_calls = 0;

bool file_exists(const std::string& path_in) {
    std::string p = path_in;
    if (!p.empty() && p.back() == '/') p.pop_back();   // src/utils.cpp:37-47
    struct stat st;
    return 0 == stat(p.c_str(), &st);
}

unsigned convert_to_8bit(double x) {
    // ConvertTo8Bit: NearestInt(Clamp(x, 0, 1) * 255) with NearestInt = floor(v + 0.5)
    const double c = std::min(std::max(x, 0.0), 1.0);
    return (unsigned)(int)std::floor(c * 255.0 + 0.5);
}

// ------------------------------------------------------------------ BMP ----

// The file is taken as bytes and its little-endian fields are picked by offset (BITMAPFILEHEADER at 0, BITMAPINFOHEADER at
// 14: width 18, height 22, planes 26, bits per pixel 28, palette entries 46); no packed structs are overlaid on it.
namespace {
constexpr size_t kBmpHeaderBytes = 14 + 40;

uint32_t le16(const std::vector<unsigned char>& b, size_t at) { return (uint32_t)b[at] | ((uint32_t)b[at + 1] << 8); }
int32_t le32(const std::vector<unsigned char>& b, size_t at) {
    return (int32_t)((uint32_t)b[at] | ((uint32_t)b[at + 1] << 8) | ((uint32_t)b[at + 2] << 16) | ((uint32_t)b[at + 3] << 24));
}
void put16(std::vector<unsigned char>& b, size_t at, uint32_t v) { b[at] = (unsigned char)(v & 0xff); b[at + 1] = (unsigned char)((v >> 8) & 0xff); }
void put32(std::vector<unsigned char>& b, size_t at, uint32_t v) { put16(b, at, v & 0xffffu); put16(b, at + 2, v >> 16); }

bool read_whole_file(const char* path, std::vector<unsigned char>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    unsigned char chunk[1 << 16];
    size_t n;
    while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) out.insert(out.end(), chunk, chunk + n);
    fclose(f);
    return true;
}

size_t padded_row_bytes(size_t width, size_t bytes_per_pixel) { return (width * bytes_per_pixel + 3) / 4 * 4; }

struct FileCloser {
    FILE* f;
    ~FileCloser() { if (f) fclose(f); }
};
}  // namespace

// Bitmap::LoadBMP (src/bitmap.cpp:101-179): uncompressed 8 / 24 / 32 bits per pixel, one plane; rows stored bottom-up and
// padded to four bytes; true-colour pixels are B, G, R; an 8-bit pixel indexes a palette whose entries the reference turns
// into colours with Color(unsigned) (src/color.cpp:10-14: red from the LOW byte), i.e. with red and blue swapped relative to
// the true-colour path -- kept, it only shows on colour palettes.
bool load_bmp(const char* path, Image& img, std::string& err) {
    img = Image();
    std::vector<unsigned char> file;
    if (!read_whole_file(path, file)) { err = std::string("loadBMP: can't open file `") + path + "'"; return false; }
    if (file.size() < 2 || file[0] != 'B' || file[1] != 'M') { err = std::string("loadBMP: `") + path + "' is not a BMP file"; return false; }
    if (file.size() < kBmpHeaderBytes) { err = "loadBMP: short header"; return false; }
    const int32_t pixel_offset = le32(file, 10), width = le32(file, 18), height = le32(file, 22), palette_field = le32(file, 46);
    const uint32_t planes = le16(file, 26), bits = le16(file, 28);
    if (bits != 8 && bits != 24 && bits != 32) { err = "loadBMP: cannot handle this bit depth"; return false; }
    if (planes != 1) { err = "loadBMP: cannot load multichannel .bmp"; return false; }
    if (width <= 0 || height <= 0) { err = "loadBMP: bad dimensions"; return false; }

    std::vector<float> palette;                       // 3 floats per entry, in the order the reference's Color(unsigned) yields them
    if (bits == 8) {
        const int entries = palette_field ? palette_field : 256;
        if (entries < 0 || entries > 256) { err = "loadBMP: bad palette size"; return false; }
        if (file.size() < kBmpHeaderBytes + (size_t)entries * 4) { err = "loadBMP: short palette"; return false; }
        palette.assign(256 * 3, 0.f);
        for (int e = 0; e < entries; ++e)
            for (int c = 0; c < 3; ++c) palette[(size_t)e * 3 + c] = file[kBmpHeaderBytes + (size_t)e * 4 + c] / 255.f;
    }
    const size_t bpp = bits / 8, stride = padded_row_bytes((size_t)width, bpp);
    if (pixel_offset < 0 || file.size() < (size_t)pixel_offset + stride * (size_t)height) {
        err = std::string("loadBMP: short read in `") + path + "'";
        return false;
    }
    img.make((uint32_t)width, (uint32_t)height);
    for (int32_t file_row = 0; file_row < height; ++file_row) {
        const unsigned char* px = file.data() + (size_t)pixel_offset + stride * (size_t)file_row;
        const int y = height - 1 - file_row;          // the first stored row is the bottom one
        for (int32_t x = 0; x < width; ++x, px += bpp) {
            if (bits == 8) img.set(x, y, palette[(size_t)px[0] * 3], palette[(size_t)px[0] * 3 + 1], palette[(size_t)px[0] * 3 + 2]);
            else img.set(x, y, px[2] / 255.0f, px[1] / 255.0f, px[0] / 255.0f);
        }
    }
    return true;
}

// Bitmap::SaveBMP (src/bitmap.cpp:181-220): 24 bits per pixel, rows bottom-up, B, G, R through ConvertTo8Bit (never sRGB).
bool save_bmp(const char* path, const float* rgb, int w, int h, size_t pitch, std::string& err) {
    const size_t stride = padded_row_bytes((size_t)w, 3);
    std::vector<unsigned char> file(kBmpHeaderBytes + stride * (size_t)h, 0);
    file[0] = 'B'; file[1] = 'M';
    put32(file, 2, (uint32_t)file.size());            // file size
    put32(file, 10, (uint32_t)kBmpHeaderBytes);       // where the pixels start
    put32(file, 14, 40);                              // BITMAPINFOHEADER
    put32(file, 18, (uint32_t)w); put32(file, 22, (uint32_t)h);
    put16(file, 26, 1); put16(file, 28, 24);          // one plane, 24 bits; compression, sizes, resolution, palette: zero
    for (int y = 0; y < h; ++y) {
        unsigned char* out = file.data() + kBmpHeaderBytes + stride * (size_t)(h - 1 - y);
        const float* in = rgb + (size_t)y * pitch;
        for (int x = 0; x < w; ++x, in += 3, out += 3)
            for (int c = 0; c < 3; ++c) out[c] = (unsigned char)convert_to_8bit(in[2 - c]);
    }
    FILE* f = fopen(path, "wb");
    if (!f) { err = std::string("saveBMP: can't open `") + path + "'"; return false; }
    const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
    fclose(f);
    if (!ok) { err = std::string("saveBMP: short write to `") + path + "'"; return false; }
    return true;
}

// ------------------------------------------------------------------ EXR ----

namespace {

float half_to_float(uint16_t hbits) {
    const uint32_t s = (hbits >> 15) & 1u, e = (hbits >> 10) & 0x1fu, m = hbits & 0x3ffu;
    uint32_t out;
    if (e == 0) {
        if (m == 0) out = s << 31;
        else {  // subnormal half -> normal float
            int ee = -1; uint32_t mm = m;
            do { ++ee; mm <<= 1; } while (!(mm & 0x400u));
            out = (s << 31) | ((uint32_t)(127 - 15 - ee) << 23) | ((mm & 0x3ffu) << 13);
        }
    } else if (e == 31) {
        out = (s << 31) | 0x7f800000u | (m << 13);
    } else {
        out = (s << 31) | ((e + 127 - 15) << 23) | (m << 13);
    }
    float f; memcpy(&f, &out, 4);
    return f;
}

// --- PIZ: canonical Huffman over 16-bit symbols ----------------------------
const int kHufEncSize = (1 << 16) + 1;

struct BitReader {
    const unsigned char* p; const unsigned char* end;
    uint64_t c = 0; int lc = 0;
    bool fail = false;
    unsigned get(int n) {
        while (lc < n) {
            if (p >= end) { fail = true; return 0; }
            c = (c << 8) | *p++; lc += 8;
        }
        lc -= n;
        return (unsigned)((c >> lc) & ((1u << n) - 1));
    }
};

bool huf_uncompress(const unsigned char* in, size_t n_in, uint16_t* out, size_t n_out, std::string& err) {
    if (n_in < 20) { err = "EXR/PIZ: short huffman block"; return false; }
    auto rd32 = [&](size_t o) { uint32_t v; memcpy(&v, in + o, 4); return v; };
    const uint32_t im = rd32(0), iM = rd32(4), n_bits = rd32(12);
    if (im >= (uint32_t)kHufEncSize || iM >= (uint32_t)kHufEncSize || im > iM) { err = "EXR/PIZ: bad huffman range"; return false; }

    // code lengths, 6 bits each with zero-run escapes
    std::vector<uint8_t> len((size_t)kHufEncSize, 0);
    BitReader br{in + 20, in + n_in};
    for (uint32_t i = im; i <= iM; ++i) {
        const unsigned l = br.get(6);
        if (br.fail) { err = "EXR/PIZ: truncated code table"; return false; }
        if (l == 63) {
            unsigned run = br.get(8) + 6;
            if (i + run > iM + 1) { err = "EXR/PIZ: code table overrun"; return false; }
            i += run - 1;   // lengths already zero
        } else if (l >= 59) {
            unsigned run = l - 59 + 2;
            if (i + run > iM + 1) { err = "EXR/PIZ: code table overrun"; return false; }
            i += run - 1;
        } else {
            len[i] = (uint8_t)l;
        }
    }
    const unsigned char* data = br.p;   // bit stream restarts on the next byte

    // canonical codes: longest codes get the smallest values
    uint64_t count[59] = {0}, base[59] = {0};
    for (uint32_t i = im; i <= iM; ++i) count[len[i]]++;
    count[0] = 0;
    {
        uint64_t c = 0;
        for (int l = 58; l > 0; --l) { const uint64_t nc = (c + count[l]) >> 1; base[l] = c; c = nc; }
    }
    // symbols of each length in increasing symbol order
    size_t first[60] = {0};
    for (int l = 1; l <= 58; ++l) first[l + 1] = first[l] + count[l];
    std::vector<uint32_t> syms(first[59]);
    {
        size_t fill[60]; memcpy(fill, first, sizeof fill);
        for (uint32_t i = im; i <= iM; ++i) if (len[i]) syms[fill[len[i]]++] = i;
    }

    if ((uint64_t)n_bits > 8ull * (uint64_t)(in + n_in - data)) { err = "EXR/PIZ: bad huffman bit count"; return false; }
    BitReader dr{data, in + n_in};
    size_t produced = 0;
    uint64_t bits_left = n_bits;
    while (produced < n_out && bits_left > 0) {
        uint64_t code = 0; int l = 0; int64_t sym = -1;
        while (bits_left > 0 && l < 58) {
            code = (code << 1) | dr.get(1); --bits_left; ++l;
            if (dr.fail) { err = "EXR/PIZ: truncated bit stream"; return false; }
            if (count[l] && code >= base[l] && code - base[l] < count[l]) { sym = syms[first[l] + (code - base[l])]; break; }
        }
        if (sym < 0) break;   // trailing pad bits
        if ((uint32_t)sym == iM) {   // run-length marker: repeat previous symbol
            if (bits_left < 8) { err = "EXR/PIZ: truncated run"; return false; }
            unsigned run = dr.get(8); bits_left -= 8;
            if (produced == 0 || produced + run > n_out) { err = "EXR/PIZ: bad run"; return false; }
            const uint16_t s = out[produced - 1];
            while (run--) out[produced++] = s;
        } else {
            out[produced++] = (uint16_t)sym;
        }
    }
    if (produced != n_out) { err = "EXR/PIZ: huffman output size mismatch"; return false; }
    return true;
}

// --- PIZ: 2-D wavelet -------------------------------------------------------
inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int ls = (int16_t)l, hs = (int16_t)h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    a = (uint16_t)(int16_t)ai;
    b = (uint16_t)(int16_t)(ai - hs);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xFFFF;
    const int aa = (d + bb - 0x8000) & 0xFFFF;
    b = (uint16_t)bb; a = (uint16_t)aa;
}

void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1;
    while (p <= n) p <<= 1;
    p >>= 1;
    int p2 = p;
    p >>= 1;
    auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { if (w14) wdec14(l, h, a, b); else wdec16(l, h, a, b); };
    while (p >= 1) {
        uint16_t* py = in;
        uint16_t* ey = in + oy * (ny - p2);
        const int oy1 = oy * p, oy2 = oy * p2, ox1 = ox * p, ox2 = ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t* px = py;
            uint16_t* ex = py + ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1; uint16_t* p10 = px + oy1; uint16_t* p11 = p10 + ox1;
                dec(*px, *p10, i00, i10);
                dec(*p01, *p11, i01, i11);
                dec(i00, i01, *px, *p01);
                dec(i10, i11, *p10, *p11);
            }
            if (nx & p) {
                uint16_t* p10 = px + oy1;
                dec(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {
            uint16_t* px = py;
            uint16_t* ex = py + ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1;
                dec(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}

struct ExrChannel { std::string name; int type; int xs, ys; };

}  // namespace

bool load_exr(const char* path, Image& img, std::string& err) {
    img = Image();
    FILE* fp = fopen(path, "rb");
    if (!fp) { err = std::string("loadEXR: can't open `") + path + "'"; return false; }
    std::vector<unsigned char> buf;
    {
        FileCloser closer{fp};
        fseek(fp, 0, SEEK_END); const long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
        if (sz < 8) { err = "loadEXR: short file"; return false; }
        buf.resize((size_t)sz);
        if (fread(buf.data(), 1, (size_t)sz, fp) != (size_t)sz) { err = "loadEXR: short read"; return false; }
    }
    size_t pos = 0;
    auto need = [&](size_t n) { return pos + n <= buf.size(); };
    auto rd_i32 = [&](size_t o) { int32_t v; memcpy(&v, &buf[o], 4); return v; };
    if (rd_i32(0) != 0x01312f76) { err = "loadEXR: bad magic"; return false; }
    const uint32_t version = (uint32_t)rd_i32(4);
    if (version & 0x1e00u) { err = "loadEXR: tiled / deep / multi-part files are not supported"; return false; }
    pos = 8;
    std::vector<ExrChannel> chans;
    int compression = -1, line_order = 0;
    int dw[4] = {0, 0, -1, -1};
    while (true) {
        auto rd_str = [&](std::string& s) {
            s.clear();
            while (pos < buf.size() && buf[pos]) s.push_back((char)buf[pos++]);
            if (pos >= buf.size()) return false;
            ++pos; return true;
        };
        std::string name, type;
        if (!rd_str(name)) { err = "loadEXR: truncated header"; return false; }
        if (name.empty()) break;
        if (!rd_str(type) || !need(4)) { err = "loadEXR: truncated header"; return false; }
        const int32_t size = rd_i32(pos); pos += 4;
        if (size < 0 || !need((size_t)size)) { err = "loadEXR: truncated attribute"; return false; }
        const size_t a0 = pos;
        const size_t a1 = a0 + (size_t)size;                  // end of this attribute (inside the file: checked above)
        if (name == "channels") {
            size_t q = a0;
            while (q < a1 && buf[q]) {
                ExrChannel c;
                while (q < a1 && buf[q]) c.name.push_back((char)buf[q++]);
                if (q >= a1 || a1 - (q + 1) < 16) { err = "loadEXR: truncated channel list"; return false; }
                ++q;
                c.type = rd_i32(q); c.xs = rd_i32(q + 8); c.ys = rd_i32(q + 12);
                q += 16;
                chans.push_back(c);
            }
        } else if (name == "compression") { if (size < 1) { err = "loadEXR: bad compression attribute"; return false; } compression = buf[a0]; }
        else if (name == "dataWindow") { if (size < 16) { err = "loadEXR: bad dataWindow attribute"; return false; } for (int i = 0; i < 4; ++i) dw[i] = rd_i32(a0 + 4 * i); }
        else if (name == "lineOrder") { if (size < 1) { err = "loadEXR: bad lineOrder attribute"; return false; } line_order = buf[a0]; }
        pos = a0 + (size_t)size;
    }
    (void)line_order;
    const int W = dw[2] - dw[0] + 1, H = dw[3] - dw[1] + 1;
    if (W <= 0 || H <= 0 || chans.empty()) { err = "loadEXR: bad header"; return false; }
    if (compression != 0 && compression != 4) { err = "loadEXR: only uncompressed and PIZ scan-line files are supported"; return false; }
    for (const ExrChannel& c : chans)
        if (c.type != 1 || c.xs != 1 || c.ys != 1) { err = "loadEXR: only HALF channels with 1x1 sampling are supported"; return false; }
    const int lines_per_chunk = compression == 4 ? 32 : 1;
    const int n_chunks = (H + lines_per_chunk - 1) / lines_per_chunk;
    const int nch = (int)chans.size();
    if (!need((size_t)n_chunks * 8)) { err = "loadEXR: truncated offset table"; return false; }
    std::vector<uint64_t> offsets((size_t)n_chunks);
    memcpy(offsets.data(), &buf[pos], (size_t)n_chunks * 8);

    int ci_r = -1, ci_g = -1, ci_b = -1;
    for (int i = 0; i < nch; ++i) {
        if (chans[i].name == "R") ci_r = i;
        else if (chans[i].name == "G") ci_g = i;
        else if (chans[i].name == "B") ci_b = i;
    }
    img.make((uint32_t)W, (uint32_t)H);

    std::vector<uint16_t> tmp, lines;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const size_t o = (size_t)offsets[ch];
        if (o + 8 > buf.size()) { err = "loadEXR: bad chunk offset"; return false; }
        const int y0 = rd_i32(o) - dw[1];
        const int data_size = rd_i32(o + 4);
        if (data_size < 0 || o + 8 + (size_t)data_size > buf.size() || y0 < 0 || y0 >= H) { err = "loadEXR: bad chunk"; return false; }
        const int ny = std::min(lines_per_chunk, H - y0);
        const size_t n_raw = (size_t)ny * nch * W;
        const unsigned char* d = &buf[o + 8];
        lines.assign(n_raw, 0);
        if ((size_t)data_size == n_raw * 2) {
            memcpy(lines.data(), d, n_raw * 2);          // stored raw
        } else {
            if (compression != 4 || data_size < 8) { err = "loadEXR: bad chunk size"; return false; }
            uint16_t min_nz, max_nz;
            memcpy(&min_nz, d, 2); memcpy(&max_nz, d + 2, 2);
            std::vector<unsigned char> bitmap(8192, 0);
            size_t q = 4;
            if (min_nz <= max_nz) {
                const size_t nb = (size_t)max_nz - min_nz + 1;
                if (max_nz >= 8192 || q + nb > (size_t)data_size) { err = "loadEXR: bad PIZ bitmap"; return false; }
                memcpy(&bitmap[min_nz], d + q, nb); q += nb;
            }
            std::vector<uint16_t> lut(65536, 0);
            int k = 0;
            for (int i = 0; i < 65536; ++i)
                if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t)i;
            const uint16_t max_value = (uint16_t)(k - 1);
            if (q + 4 > (size_t)data_size) { err = "loadEXR: bad PIZ block"; return false; }
            const int32_t huf_len = rd_i32(o + 8 + q); q += 4;
            if (huf_len < 0 || q + (size_t)huf_len > (size_t)data_size) { err = "loadEXR: bad PIZ huffman length"; return false; }
            tmp.assign(n_raw, 0);
            if (!huf_uncompress(d + q, (size_t)huf_len, tmp.data(), n_raw, err)) return false;
            for (int c = 0; c < nch; ++c)
                wav2_decode(tmp.data() + (size_t)c * ny * W, W, 1, ny, W, max_value);
            for (size_t i = 0; i < n_raw; ++i) tmp[i] = lut[tmp[i]];
            // channel-planar block -> scan-line interleaved (per line: channel 0 row, channel 1 row, ...)
            for (int y = 0; y < ny; ++y)
                for (int c = 0; c < nch; ++c)
                    memcpy(&lines[((size_t)y * nch + c) * W], &tmp[((size_t)c * ny + y) * W], (size_t)W * 2);
        }
        for (int y = 0; y < ny; ++y) {
            const uint16_t* row = &lines[(size_t)y * nch * W];
            for (int x = 0; x < W; ++x) {
                const float r = ci_r >= 0 ? half_to_float(row[(size_t)ci_r * W + x]) : 0.f;
                const float g = ci_g >= 0 ? half_to_float(row[(size_t)ci_g * W + x]) : 0.f;
                const float b = ci_b >= 0 ? half_to_float(row[(size_t)ci_b * W + x]) : 0.f;
                img.set((uint32_t)x, (uint32_t)(y0 + y), r, g, b);
            }
        }
    }
    return true;
}

static std::string extension_upper(const char* path) {
    const char* dot = strrchr(path, '.');
    std::string e;
    if (!dot || strlen(path) < 2) return e;
    for (const char* p = dot + 1; *p; ++p) e.push_back((char)toupper((unsigned char)*p));
    return e;
}

// float -> half as OpenEXR's half(float) does it: round to nearest even, overflow to infinity, small values to subnormals
uint16_t float_to_half(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ex = (x >> 23) & 0xffu;
    uint32_t man = x & 0x7fffffu;
    if (ex == 0xffu) return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0u));      // inf / nan
    const int e = (int)ex - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;                      // below half of the smallest subnormal
        man |= 0x800000u;
        const int shift = 14 - e;                                // 14 .. 24
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1u), half_way = 1u << (shift - 1);
        if (rem > half_way || (rem == half_way && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;      // a carry into the exponent is the right result, up to infinity
    return (uint16_t)(sign | h);
}

// Bitmap::SaveEXR (src/bitmap.cpp:254-277): RGBA half, alpha 1.  The reference goes through Imf::RgbaOutputFile (PIZ
// chunks); this writer emits the same image as an uncompressed scan-line file (OpenEXR file layout: magic, version 2,
// attributes, offset table, one chunk per line with the channels A, B, G, R in that order), which any OpenEXR reader decodes
// to the same half pixels.
bool save_exr(const char* path, const float* rgb, int w, int h, size_t pitch, std::string& err) {
    if (w <= 0 || h <= 0) { err = "saveEXR: empty image"; return false; }
    FILE* fp = fopen(path, "wb");
    if (!fp) { err = std::string("saveEXR: can't open `") + path + "'"; return false; }
    FileCloser closer{fp};
    std::vector<unsigned char> hd;
    auto put = [&](const void* p, size_t n) { hd.insert(hd.end(), (const unsigned char*)p, (const unsigned char*)p + n); };
    auto put_str = [&](const char* z) { put(z, strlen(z) + 1); };
    auto put_i32 = [&](int32_t v) { put(&v, 4); };
    auto put_f32 = [&](float v) { put(&v, 4); };
    auto attr = [&](const char* name, const char* type, int32_t size) { put_str(name); put_str(type); put_i32(size); };
    put_i32(20000630); put_i32(2);
    attr("channels", "chlist", 4 * 18 + 1);
    for (const char* ch : {"A", "B", "G", "R"}) { put_str(ch); put_i32(1); put_i32(0); put_i32(1); put_i32(1); }    // HALF, pLinear 0, 1x1
    hd.push_back(0);
    attr("compression", "compression", 1); hd.push_back(0);
    for (const char* name : {"dataWindow", "displayWindow"}) { attr(name, "box2i", 16); put_i32(0); put_i32(0); put_i32(w - 1); put_i32(h - 1); }
    attr("lineOrder", "lineOrder", 1); hd.push_back(0);
    attr("pixelAspectRatio", "float", 4); put_f32(1.f);
    attr("screenWindowCenter", "v2f", 8); put_f32(0.f); put_f32(0.f);
    attr("screenWindowWidth", "float", 4); put_f32(1.f);
    hd.push_back(0);
    const uint64_t line_bytes = (uint64_t)w * 4 * 2, chunk = 8 + line_bytes;
    const uint64_t first = hd.size() + (uint64_t)h * 8;
    for (int y = 0; y < h; ++y) { const uint64_t off = first + (uint64_t)y * chunk; put(&off, 8); }
    if (fwrite(hd.data(), hd.size(), 1, fp) != 1) { err = "saveEXR: write failed"; return false; }
    std::vector<uint16_t> line((size_t)w * 4);
    const uint16_t one = float_to_half(1.f);
    for (int y = 0; y < h; ++y) {
        const float* src = rgb + (size_t)y * pitch;
        for (int x = 0; x < w; ++x) {
            line[x] = one;
            line[(size_t)w + x] = float_to_half(src[x * 3 + 2]);
            line[(size_t)2 * w + x] = float_to_half(src[x * 3 + 1]);
            line[(size_t)3 * w + x] = float_to_half(src[x * 3 + 0]);
        }
        const int32_t yy = y, size = (int32_t)line_bytes;
        if (fwrite(&yy, 4, 1, fp) != 1 || fwrite(&size, 4, 1, fp) != 1 || fwrite(line.data(), (size_t)line_bytes, 1, fp) != 1) { err = "saveEXR: write failed"; return false; }
    }
    return true;
}

bool load_image(const char* path, Image& img, std::string& err) {
    const std::string e = extension_upper(path);
    if (e == "BMP") return load_bmp(path, img, err);
    if (e == "EXR") return load_exr(path, img, err);
    err = std::string("unsupported image type: ") + path;
    return false;
}

void differentiate(Image& img) {
    if (g_differentiate_hook && img.w && img.h && g_differentiate_hook(img.w, img.h, img.rgb.data()) == 0) { ++g_differentiate_hook_calls; return; }   // device pass; on failure the CPU code below
    // (I(x,y) - I(x+1,y), I(x,y) - I(x,y+1), 0) with wrap-around; I = plain mean (src/color.h:18)
    Image out; out.make(img.w, img.h);
    auto inten = [&](uint32_t x, uint32_t y) { float p[3]; img.get(x, y, p); return (p[0] + p[1] + p[2]) / 3; };
    for (uint32_t y = 0; y < img.h; ++y)
        for (uint32_t x = 0; x < img.w; ++x) {
            const float dx = inten(x, y) - inten((x + 1) % img.w, y);
            const float dy = inten(x, y) - inten(x, (y + 1) % img.h);
            out.set(x, y, dx, dy, 0.f);
        }
    img.rgb.swap(out.rgb);
}

}  // namespace qdh
