// png_codec.cpp — minimal PNG reader/writer + PPM writer for the host front-end (no external library).
//
// Reader: what Texture2d::from_png needs from stb_image (src/texture2d.rs:78-177): 8-bit-per-channel
// images of 1, 2, 3 or 4 channels, top row first.  Supports colour types 0/2/3/4/6, bit depths
// 1-16 (16 keeps the high byte, <8 is expanded), Adam7 interlacing, zlib stored/fixed/dynamic blocks.
// The other formats stb_image decodes (TGA, BMP, JPEG) and the format dispatch are in image_codec.cpp.
// Writer: Image::to_png (src/image.rs:60-90) — RGB8 with `clamp(c*255, 0, 255)` truncated to u8 —
// and Image::to_ppm (src/image.rs:27-58).
#include "host.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace nrays_host {
namespace {

uint32_t crc_table[256];
bool crc_ready = false;
void crc_init() {
    for (uint32_t n = 0; n < 256; ++n) { uint32_t c = n; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; }
    crc_ready = true;
}
uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0xffffffffu) {
    if (!crc_ready) crc_init();
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return crc;
}
uint32_t adler32(const uint8_t* p, size_t n) {
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; ++i) { a = (a + p[i]) % 65521u; b = (b + a) % 65521u; }
    return (b << 16) | a;
}

// ---- inflate (RFC 1951) -----------------------------------------------------------------------
struct Bits {
    const uint8_t* p; size_t n, pos = 0; uint32_t buf = 0; int cnt = 0;
    int get(int k) {
        while (cnt < k) { if (pos >= n) throw std::runtime_error("png: truncated zlib stream"); buf |= (uint32_t)p[pos++] << cnt; cnt += 8; }
        int v = (int)(buf & ((1u << k) - 1)); buf >>= k; cnt -= k; return v;
    }
    void align() { buf = 0; cnt = 0; }
};
struct Huff {
    uint16_t count[16] = {0}; uint16_t symbol[288];
    void build(const uint8_t* len, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16]; offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(Bits& b) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; ++l) {
            code |= b.get(1);
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        throw std::runtime_error("png: bad huffman code");
    }
};
const uint16_t kLenBase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
const uint16_t kLenExtra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
const uint16_t kDistBase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
const uint16_t kDistExtra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

std::vector<uint8_t> inflate_zlib(const std::vector<uint8_t>& z) {
    if (z.size() < 6) throw std::runtime_error("png: zlib stream too short");
    Bits b{z.data() + 2, z.size() - 2};
    std::vector<uint8_t> out;
    for (;;) {
        int final = b.get(1), type = b.get(2);
        if (type == 0) {
            b.align();
            if (b.pos + 4 > b.n) throw std::runtime_error("png: truncated stored block");
            uint32_t len = b.p[b.pos] | (b.p[b.pos + 1] << 8); b.pos += 4;
            if (b.pos + len > b.n) throw std::runtime_error("png: truncated stored block");
            out.insert(out.end(), b.p + b.pos, b.p + b.pos + len); b.pos += len;
        } else if (type == 1 || type == 2) {
            Huff lit, dist;
            if (type == 1) {
                uint8_t l[288]; for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                lit.build(l, 288); uint8_t d[30]; std::memset(d, 5, 30); dist.build(d, 30);
            } else {
                int nlen = b.get(5) + 257, ndist = b.get(5) + 1, ncode = b.get(4) + 4;
                static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                uint8_t cl[19] = {0}; for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)b.get(3);
                Huff ch; ch.build(cl, 19);
                uint8_t lens[320]; int i = 0;
                while (i < nlen + ndist) {
                    int s = ch.decode(b);
                    if (s < 16) lens[i++] = (uint8_t)s;
                    else {
                        int rep, val = 0;
                        if (s == 16) { if (!i) throw std::runtime_error("png: bad repeat"); val = lens[i - 1]; rep = 3 + b.get(2); }
                        else if (s == 17) rep = 3 + b.get(3); else rep = 11 + b.get(7);
                        if (i + rep > nlen + ndist) throw std::runtime_error("png: bad code lengths");
                        while (rep--) lens[i++] = (uint8_t)val;
                    }
                }
                lit.build(lens, nlen); dist.build(lens + nlen, ndist);
            }
            for (;;) {
                int s = lit.decode(b);
                if (s < 256) out.push_back((uint8_t)s);
                else if (s == 256) break;
                else {
                    s -= 257; if (s >= 29) throw std::runtime_error("png: bad length symbol");
                    int len = kLenBase[s] + b.get(kLenExtra[s]);
                    int ds = dist.decode(b); if (ds >= 30) throw std::runtime_error("png: bad distance symbol");
                    size_t d = kDistBase[ds] + b.get(kDistExtra[ds]);
                    if (d > out.size()) throw std::runtime_error("png: distance too far back");
                    size_t from = out.size() - d;
                    for (int k = 0; k < len; ++k) out.push_back(out[from + k]);
                }
            }
        } else throw std::runtime_error("png: bad block type");
        if (final) break;
    }
    return out;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
int paeth(int a, int b, int c) { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

} // namespace

Image8 read_png(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Image not found: " + path);
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8)) throw std::runtime_error("not a PNG file: " + path);
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    for (size_t p = 8; p + 12 <= d.size();) {
        uint32_t len = be32(&d[p]); std::string type((const char*)&d[p + 4], 4);
        if (p + 12 + len > d.size()) throw std::runtime_error("png: truncated chunk");
        const uint8_t* c = &d[p + 8];
        if (type == "IHDR") {
            if (len != 13) throw std::runtime_error("png: bad IHDR length: " + path);
            w = be32(c); h = be32(c + 4); depth = c[8]; ctype = c[9]; interlace = c[12];
        }
        else if (type == "PLTE") plte.assign(c, c + len);
        else if (type == "tRNS") trns.assign(c, c + len);
        else if (type == "IDAT") idat.insert(idat.end(), c, c + len);
        else if (type == "IEND") break;
        p += 12 + len;
    }
    if (!w || !h) throw std::runtime_error("png: empty image: " + path);
    if (interlace > 1) throw std::runtime_error("png: bad interlace method: " + path);
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) throw std::runtime_error("png: bad colour type");
    // legal (colour type, bit depth) pairs of the PNG specification; anything else would give a zero or garbage stride
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                     : (depth == 8 || depth == 16);
    if (!depth_ok) throw std::runtime_error("png: bad bit depth for the colour type: " + path);
    if ((uint64_t)w * h > (1ull << 28)) throw std::runtime_error("png: image too large: " + path);
    const size_t bpp_bits = (size_t)ch * depth, bpp = std::max<size_t>(1, bpp_bits / 8);
    std::vector<uint8_t> raw = inflate_zlib(idat);
    std::vector<uint8_t> s8((size_t)w * h * ch);
    // One pass = a sub-image of pw x ph pixels whose pixel (i, j) is pixel (x0 + i dx, y0 + j dy) of the image: the whole image for
    // a non-interlaced file, the seven Adam7 passes otherwise (stb_image decodes both).  Scanlines are unfiltered per pass.
    size_t rpos = 0;
    auto pass = [&](uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy) {
        if (x0 >= w || y0 >= h) return;
        const uint32_t pw = (w - x0 + dx - 1) / dx, ph = (h - y0 + dy - 1) / dy;
        const size_t stride = (pw * bpp_bits + 7) / 8;
        if (raw.size() < rpos + (stride + 1) * ph) throw std::runtime_error("png: short image data");
        std::vector<uint8_t> img(stride * ph), zero(stride, 0);
        for (uint32_t y = 0; y < ph; ++y) {
            const uint8_t* s = &raw[rpos + (stride + 1) * y]; int ft = s[0]; ++s;
            if (ft > 4) throw std::runtime_error("png: bad filter type");
            uint8_t* o = &img[stride * y]; const uint8_t* up = y ? &img[stride * (y - 1)] : zero.data();
            for (size_t x = 0; x < stride; ++x) {
                int a = x >= bpp ? o[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0, v = s[x];
                switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break; case 4: v += paeth(a, b, c); break; default: break; }
                o[x] = (uint8_t)v;
            }
        }
        rpos += (stride + 1) * ph;
        // unpack to 8 bits per sample
        for (uint32_t y = 0; y < ph; ++y) for (uint32_t px = 0; px < pw; ++px) for (int cc = 0; cc < ch; ++cc) {
            const uint32_t x = px * (uint32_t)ch + (uint32_t)cc;
            const uint8_t* row = &img[stride * y]; uint8_t v;
            if (depth == 8) v = row[x];
            else if (depth == 16) v = row[2 * x];
            else { int per = 8 / depth, sh = (per - 1 - (int)(x % per)) * depth; int raw_v = (row[x / per] >> sh) & ((1 << depth) - 1);
                   v = ctype == 3 ? (uint8_t)raw_v : (uint8_t)(raw_v * 255 / ((1 << depth) - 1)); }
            s8[((size_t)(y0 + y * dy) * w + (x0 + px * dx)) * ch + cc] = v;
        }
    };
    if (!interlace) pass(0, 0, 1, 1);
    else {
        static const uint32_t a7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
        for (const auto& q : a7) pass(q[0], q[1], q[2], q[3]);
    }
    Image8 out; out.width = w; out.height = h;
    if (ctype == 3) { // palette -> RGB (RGBA if tRNS), like stb_image
        out.channels = trns.empty() ? 3 : 4;
        out.data.resize((size_t)w * h * out.channels);
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            size_t k = s8[i];
            for (int c = 0; c < 3; ++c) out.data[i * out.channels + c] = 3 * k + c < plte.size() ? plte[3 * k + c] : 0;
            if (out.channels == 4) out.data[i * 4 + 3] = k < trns.size() ? trns[k] : 255;
        }
    } else { out.channels = ch; out.data.swap(s8); }
    return out;
}

void write_png_rgb8(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw; raw.reserve((size_t)(3 * w + 1) * h);
    for (uint32_t y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), rgb + (size_t)3 * w * y, rgb + (size_t)3 * w * (y + 1)); }
    std::vector<uint8_t> z = {0x78, 0x01};
    for (size_t p = 0; p < raw.size() || p == 0; p += 65535) { // stored deflate blocks
        size_t n = std::min<size_t>(65535, raw.size() - p);
        z.push_back(p + n >= raw.size() ? 1 : 0); z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + p, raw.begin() + p + n);
        if (raw.empty()) break;
    }
    put32(z, adler32(raw.data(), raw.size()));
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto chunk = [&](const char* type, const std::vector<uint8_t>& body) {
        put32(out, (uint32_t)body.size());
        std::vector<uint8_t> t(type, type + 4); t.insert(t.end(), body.begin(), body.end());
        out.insert(out.end(), t.begin(), t.end()); put32(out, crc32(t.data(), t.size()) ^ 0xffffffffu);
    };
    std::vector<uint8_t> ihdr; put32(ihdr, w); put32(ihdr, h); ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    chunk("IHDR", ihdr); chunk("IDAT", z); chunk("IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Failed to save the output image: " + path);
    f.write((const char*)out.data(), (std::streamsize)out.size());
}

// Image::to_png quantisation (src/image.rs:66-76): c*255, clamped to [0,255], truncated.
std::vector<uint8_t> quantize_rgb8(const float* rgb, size_t n) {
    std::vector<uint8_t> q(n);
    for (size_t i = 0; i < n; ++i) { float v = rgb[i] * 255.0f; v = (v > 0.0f) ? v : 0.0f /* NaN and negatives -> 0, as Rust's saturating `as` casts */; v = v > 255.0f ? 255.0f : v; q[i] = (uint8_t)(size_t)v; }
    return q;
}

void write_ppm(const std::string& path, const float* rgb, uint32_t w, uint32_t h) { // src/image.rs:27-58 (with the clamp fixed)
    std::vector<uint8_t> q = quantize_rgb8(rgb, (size_t)w * h * 3);
    write_ppm_rgb8(path, q.data(), w, h);
}
void write_ppm_rgb8(const std::string& path, const uint8_t* q, uint32_t w, uint32_t h) {
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) throw std::runtime_error("cannot write " + path);
    std::fprintf(f, "P3\n%u %u\n255\n", w, h);
    for (uint32_t y = 0; y < h; ++y) { for (uint32_t x = 0; x < w * 3; ++x) std::fprintf(f, "%u ", q[(size_t)y * w * 3 + x]); std::fprintf(f, "\n"); }
    std::fclose(f);
}

} // namespace nrays_host
