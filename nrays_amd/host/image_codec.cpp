// image_codec.cpp — the other file formats Texture2d::from_png accepts.  Despite its name the reference decodes every texture
// through stb_image's `image::load` (src/texture2d.rs:95), which recognises the format from the file's CONTENT: PNG, JPEG,
// BMP and — tried last, it has no signature — TGA (the format of the Crytek Sponza distribution's textures).  read_image()
// is that dispatch; the result is what stb_image hands back: 8-bit interleaved channels, top row first, channel count as
// stored in the file (1 = grey, 2 = grey + alpha, 3 = RGB, 4 = RGBA), which src/texture2d.rs:109-177 then expands.
//
//   TGA   image types 1 / 2 / 3 (colour-mapped, true colour, grey) and their run-length forms 9 / 10 / 11; 8, 15 / 16, 24 and
//         32 bits per pixel; both origins (descriptor bit 5); BGR(A) -> RGB(A), 5-5-5 -> RGB.
//   BMP   uncompressed 8-bit palette, 24 and 32 bits per pixel, bottom-up and top-down.
//   JPEG  Huffman-coded 8-bit baseline / extended-sequential (SOF0 / SOF1) and progressive (SOF2) files, grey or YCbCr, any
//         sampling factors up to 4, restart intervals.  The inverse DCT is the published IJG "islow" integer transform with
//         12-bit constants, chroma is upsampled with the triangle filters stb_image uses (3:1 along a subsampled axis), and
//         YCbCr -> RGB is the JFIF matrix in 20-bit fixed point; the crate revision the reference pins is not recorded
//         (Cargo.toml:18-22 names a git HEAD), so the last bit of a JPEG texel is not pinned — PNG, TGA and BMP are exact.
#include "host.hpp"

#include <cstring>
#include <fstream>

namespace nrays_host {
namespace {

std::vector<uint8_t> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Image not found: " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
uint32_t le16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// ------------------------------------------------------------------------------------------------ TGA
bool tga_header_ok(const std::vector<uint8_t>& d) {
    if (d.size() < 18) return false;
    const int cmap = d[1], type = d[2], bpp = d[16];
    if (cmap > 1) return false;
    if (cmap == 1) {
        if (type != 1 && type != 9) return false;
        const int eb = d[7];
        if (eb != 8 && eb != 15 && eb != 16 && eb != 24 && eb != 32) return false;
        if (bpp != 8 && bpp != 16) return false;
    } else {
        if (type != 2 && type != 3 && type != 10 && type != 11) return false;
        if (bpp != 8 && bpp != 15 && bpp != 16 && bpp != 24 && bpp != 32) return false;
    }
    return le16(&d[12]) >= 1 && le16(&d[14]) >= 1;
}

// One stored pixel (`bits` per pixel, little endian) -> `comp` output channels.
void tga_pixel(const uint8_t* p, int bits, bool grey, uint8_t* o) {
    if (bits == 8) { o[0] = p[0]; return; }
    if (bits == 15 || bits == 16) {
        if (grey) { o[0] = p[0]; o[1] = p[1]; return; }  // grey + alpha
        const uint32_t px = le16(p);                      // A RRRRR GGGGG BBBBB (the attribute bit is ignored, as stb_image does)
        const int r = (px >> 10) & 31, g = (px >> 5) & 31, b = px & 31;
        o[0] = (uint8_t)((r * 255) / 31); o[1] = (uint8_t)((g * 255) / 31); o[2] = (uint8_t)((b * 255) / 31);
        return;
    }
    o[0] = p[2]; o[1] = p[1]; o[2] = p[0];               // BGR(A) -> RGB(A)
    if (bits == 32) o[3] = p[3];
}

Image8 read_tga(const std::vector<uint8_t>& d, const std::string& path) {
    if (!tga_header_ok(d)) throw std::runtime_error("unknown image type (not PNG / JPEG / BMP / TGA): " + path);
    const int idlen = d[0], cmap = d[1], type = d[2] & 7, rle = d[2] & 8, bpp = d[16], desc = d[17];
    const uint32_t cm_first = le16(&d[3]), cm_len = le16(&d[5]); const int cm_bits = d[7];
    const uint32_t w = le16(&d[12]), h = le16(&d[14]);
    const bool grey = type == 3;
    const int src_bits = cmap ? cm_bits : bpp;  // bits of a colour value
    const int comp = src_bits == 8 ? 1 : (src_bits == 15 || src_bits == 16) ? (grey ? 2 : 3) : src_bits / 8;
    size_t pos = 18 + (size_t)idlen;
    std::vector<uint8_t> palette;
    if (cmap) {
        if (cm_len == 0) throw std::runtime_error("tga: colour-mapped image without a colour map: " + path);
        const size_t eb = (size_t)(cm_bits + 7) / 8;
        if (pos + eb * cm_len > d.size()) throw std::runtime_error("tga: truncated colour map: " + path);
        palette.resize((size_t)cm_len * comp);
        for (uint32_t i = 0; i < cm_len; ++i) tga_pixel(&d[pos + eb * i], cm_bits, false, &palette[(size_t)i * comp]);
        pos += eb * cm_len;
    } else if (cm_len) pos += (size_t)((cm_bits + 7) / 8) * cm_len; // a colour map an unmapped image does not use
    const size_t pb = (size_t)(bpp + 7) / 8;  // stored bytes per pixel (the index for colour-mapped images)
    // every pixel takes at least one stored byte (a run-length packet of 1 + pb bytes covers at most 128): a header that promises more
    // pixels than the file can hold is rejected BEFORE the output is allocated
    if ((uint64_t)w * h > (uint64_t)(d.size() - std::min(pos, d.size())) * 128u) throw std::runtime_error("tga: truncated pixel data: " + path);
    Image8 out; out.width = w; out.height = h; out.channels = comp; out.data.resize((size_t)w * h * comp);
    uint8_t px[4] = {0, 0, 0, 0};
    auto fetch = [&]() {
        if (pos + pb > d.size()) throw std::runtime_error("tga: truncated pixel data: " + path);
        if (cmap) {
            uint32_t idx = pb == 1 ? d[pos] : le16(&d[pos]);
            idx = idx >= cm_first ? idx - cm_first : 0;
            if (idx >= cm_len) idx = 0;
            std::memcpy(px, &palette[(size_t)idx * comp], (size_t)comp);
        } else tga_pixel(&d[pos], bpp, grey, px);
        pos += pb;
    };
    size_t run = 0; bool raw_run = true;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        if (rle) {
            if (run == 0) {
                if (pos >= d.size()) throw std::runtime_error("tga: truncated run: " + path);
                const int c = d[pos++];
                run = (size_t)(c & 127) + 1; raw_run = !(c & 128);
                fetch();
            } else if (raw_run) fetch();
            --run;
        } else fetch();
        std::memcpy(&out.data[i * comp], px, (size_t)comp);
    }
    if (!(desc & 0x20)) { // bottom-left origin: flip to top row first
        std::vector<uint8_t> row((size_t)w * comp);
        for (uint32_t y = 0; y < h / 2; ++y) {
            uint8_t* a = &out.data[(size_t)y * w * comp]; uint8_t* b = &out.data[(size_t)(h - 1 - y) * w * comp];
            std::memcpy(row.data(), a, row.size()); std::memcpy(a, b, row.size()); std::memcpy(b, row.data(), row.size());
        }
    }
    // (descriptor bit 4, right-to-left columns, is ignored — as stb_image ignores it)
    return out;
}

// ------------------------------------------------------------------------------------------------ BMP
Image8 read_bmp(const std::vector<uint8_t>& d, const std::string& path) {
    if (d.size() < 54) throw std::runtime_error("bmp: truncated header: " + path);
    const uint32_t off = le32(&d[10]), hsz = le32(&d[14]);
    if (hsz < 40) throw std::runtime_error("bmp: OS/2 headers are not supported: " + path);
    const int32_t w = (int32_t)le32(&d[18]), hs = (int32_t)le32(&d[22]);
    const int bpp = (int)le16(&d[28]); const uint32_t compression = le32(&d[30]);
    if (w <= 0 || hs == 0 || (uint64_t)w * (uint64_t)(hs < 0 ? -hs : hs) > (1ull << 28)) throw std::runtime_error("bmp: bad dimensions: " + path);
    if (compression != 0 && !(compression == 3 && bpp == 32)) throw std::runtime_error("bmp: compressed files are not supported: " + path);
    if (bpp != 8 && bpp != 24 && bpp != 32) throw std::runtime_error("bmp: unsupported bit depth: " + path);
    const uint32_t h = (uint32_t)(hs < 0 ? -hs : hs);
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    if ((size_t)off + stride * h > d.size()) throw std::runtime_error("bmp: truncated pixel data: " + path);
    uint32_t ncol = le32(&d[46]); if (bpp == 8 && ncol == 0) ncol = 256;
    const size_t pal = 14 + (size_t)hsz;
    if (bpp == 8 && pal + 4 * (size_t)ncol > d.size()) throw std::runtime_error("bmp: truncated palette: " + path);
    Image8 out; out.width = (uint32_t)w; out.height = h; out.channels = bpp == 32 ? 4 : 3; out.data.resize((size_t)w * h * out.channels);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = &d[off + stride * (hs < 0 ? y : h - 1 - y)];
        uint8_t* o = &out.data[(size_t)y * w * out.channels];
        for (int32_t x = 0; x < w; ++x) {
            if (bpp == 8) { const uint32_t k = row[x] < ncol ? row[x] : 0; o[3 * x] = d[pal + 4 * k + 2]; o[3 * x + 1] = d[pal + 4 * k + 1]; o[3 * x + 2] = d[pal + 4 * k]; }
            else if (bpp == 24) { o[3 * x] = row[3 * x + 2]; o[3 * x + 1] = row[3 * x + 1]; o[3 * x + 2] = row[3 * x]; }
            else { o[4 * x] = row[4 * x + 2]; o[4 * x + 1] = row[4 * x + 1]; o[4 * x + 2] = row[4 * x]; o[4 * x + 3] = compression == 3 ? row[4 * x + 3] : 255; }
        }
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ JPEG (ITU T.81)
const uint8_t kZigzag[64 + 15] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
                                  57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct JHuff {
    uint8_t bits[17] = {0}; uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    bool present = false;
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct JComp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
    uint32_t bw = 0, bh = 0;         // blocks per row / column (padded to whole MCUs)
    std::vector<int16_t> coef;       // 64 per block
    std::vector<uint8_t> plane;      // bw*8 x bh*8 samples after the inverse transform
};

struct JDec {
    const std::vector<uint8_t>& d; std::string path;
    size_t pos = 0;
    uint32_t bitbuf = 0; int bitcnt = 0; bool hit_marker = false;
    uint16_t qt[4][64]; bool qt_ok[4] = {false, false, false, false};
    JHuff hdc[4], hac[4];
    std::vector<JComp> comps;
    uint32_t width = 0, height = 0; int hmax = 1, vmax = 1; bool progressive = false;
    uint32_t restart_interval = 0, mcux = 0, mcuy = 0;
    int eobrun = 0;
    JDec(const std::vector<uint8_t>& data, const std::string& p) : d(data), path(p) {}
    [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("jpeg: ") + what + ": " + path); }

    // ---- entropy-coded segment bit reader: 0xFF00 is a stuffed 0xFF, any other marker ends the segment (zeros follow)
    void fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!hit_marker && pos < d.size()) {
                b = d[pos];
                if (b == 0xff) {
                    const int n = pos + 1 < d.size() ? d[pos + 1] : 0xd9;
                    if (n == 0) pos += 2;
                    else { hit_marker = true; b = 0; }
                } else ++pos;
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt); bitcnt += 8;
        }
    }
    int getbits(int n) { if (n == 0) return 0; if (bitcnt < n) fill(); const int v = (int)(bitbuf >> (32 - n)); bitbuf <<= n; bitcnt -= n; return v; }
    int getbit() { return getbits(1); }
    static int extend(int v, int n) { return n && v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }
    int decode(const JHuff& h) {
        if (!h.present) fail("missing Huffman table");
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = (code << 1) | getbit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        fail("bad Huffman code");
    }
    void reset_entropy() { bitbuf = 0; bitcnt = 0; hit_marker = false; eobrun = 0; for (auto& c : comps) c.dc_pred = 0; }

    // ---- one block of a sequential scan: all 64 coefficients
    void block_sequential(JComp& c, int16_t* blk) {
        const int t = decode(hdc[c.td]);
        if (t > 15) fail("bad DC size");
        c.dc_pred += extend(getbits(t), t);
        blk[0] = (int16_t)c.dc_pred;
        for (int k = 1; k < 64;) {
            const int rs = decode(hac[c.ta]), r = rs >> 4, s = rs & 15;
            if (s == 0) { if (r != 15) break; k += 16; continue; }
            k += r;
            if (k > 63) fail("bad AC run");
            blk[kZigzag[k++]] = (int16_t)extend(getbits(s), s);
        }
    }
    // ---- progressive scans (T.81 G.1.2)
    void block_dc_progressive(JComp& c, int16_t* blk, int ah, int al) {
        if (ah == 0) {
            const int t = decode(hdc[c.td]);
            if (t > 15) fail("bad DC size");
            c.dc_pred += extend(getbits(t), t);
            blk[0] = (int16_t)(c.dc_pred * (1 << al));
        } else if (getbit()) blk[0] = (int16_t)(blk[0] | (1 << al));
    }
    void block_ac_progressive(JComp& c, int16_t* blk, int ss, int se, int ah, int al) {
        if (ah == 0) {
            if (eobrun) { --eobrun; return; }
            for (int k = ss; k <= se;) {
                const int rs = decode(hac[c.ta]), r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += getbits(r); break; }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) fail("bad AC run");
                    blk[kZigzag[k++]] = (int16_t)(extend(getbits(s), s) * (1 << al));
                }
            }
            return;
        }
        const int p1 = 1 << al, m1 = -1 * (1 << al);
        auto refine = [&](int16_t& v) { if (getbit() && !(v & p1)) v = (int16_t)(v >= 0 ? v + p1 : v + m1); };
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se;) {
                const int rs = decode(hac[c.ta]); int r = rs >> 4; const int s = rs & 15;
                int val = 0;
                if (s == 0) {
                    if (r < 15) { eobrun = (1 << r); if (r) eobrun += getbits(r); break; }
                } else {
                    if (s != 1) fail("bad refinement code");
                    val = getbit() ? p1 : m1;
                }
                for (; k <= se; ++k) {
                    int16_t& v = blk[kZigzag[k]];
                    if (v != 0) refine(v);
                    else { if (r == 0) { if (val) v = (int16_t)val; ++k; break; } --r; }
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) { int16_t& v = blk[kZigzag[k]]; if (v != 0) refine(v); }
            --eobrun;
        }
    }

    // ---- markers
    uint32_t be16(size_t p) const { if (p + 2 > d.size()) fail("truncated file"); return ((uint32_t)d[p] << 8) | d[p + 1]; }
    void parse_dqt(size_t p, size_t end) {
        while (p < end) {
            const int pq = d[p] >> 4, tq = d[p] & 15; ++p;
            if (tq > 3 || pq > 1) fail("bad quantisation table");
            if (p + (pq ? 128 : 64) > end) fail("truncated quantisation table");
            for (int i = 0; i < 64; ++i) { qt[tq][kZigzag[i]] = (uint16_t)(pq ? be16(p) : d[p]); p += pq ? 2 : 1; }
            qt_ok[tq] = true;
        }
    }
    void parse_dht(size_t p, size_t end) {
        while (p < end) {
            const int tc = d[p] >> 4, th = d[p] & 15; ++p;
            if (tc > 1 || th > 3 || p + 16 > end) fail("bad Huffman table");
            JHuff& h = tc ? hac[th] : hdc[th];
            int n = 0;
            for (int l = 1; l <= 16; ++l) { h.bits[l] = d[p + l - 1]; n += h.bits[l]; }
            p += 16;
            if (n > 256 || p + n > end) fail("bad Huffman table");
            std::memcpy(h.vals, &d[p], (size_t)n); p += n;
            h.build();
        }
    }
    void parse_sof(size_t p, size_t end, bool prog) {
        if (width) fail("more than one frame");
        if (p + 6 > end || d[p] != 8) fail("only 8-bit samples are supported");
        height = be16(p + 1); width = be16(p + 3);
        const int n = d[p + 5];
        if (!width || !height) fail("empty image");
        if ((n != 1 && n != 3) || p + 6 + 3 * (size_t)n > end) fail("unsupported component count");
        progressive = prog; comps.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            JComp& c = comps[(size_t)i];
            c.id = d[p + 6 + 3 * i]; c.h = d[p + 7 + 3 * i] >> 4; c.v = d[p + 7 + 3 * i] & 15; c.tq = d[p + 8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) fail("bad sampling factors");
            hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
        }
        for (auto& c : comps) if (hmax % c.h || vmax % c.v) fail("fractional sampling ratios are not supported");
        mcux = (width + 8 * (uint32_t)hmax - 1) / (8 * (uint32_t)hmax); mcuy = (height + 8 * (uint32_t)vmax - 1) / (8 * (uint32_t)vmax);
        if ((uint64_t)mcux * mcuy * hmax * vmax > (1ull << 22)) fail("image too large");
        for (auto& c : comps) { c.bw = mcux * (uint32_t)c.h; c.bh = mcuy * (uint32_t)c.v; c.coef.assign((size_t)c.bw * c.bh * 64, 0); }
    }
    void scan(size_t p, size_t end) {
        if (!width) fail("scan before the frame header");
        if (p >= end) fail("bad scan header");
        const int ns = d[p];
        if (ns < 1 || ns > (int)comps.size() || p + 1 + 2 * (size_t)ns + 3 > end) fail("bad scan header");
        std::vector<JComp*> sc;
        for (int i = 0; i < ns; ++i) {
            const int cid = d[p + 1 + 2 * i], tabs = d[p + 2 + 2 * i];
            JComp* c = nullptr;
            for (auto& k : comps) if (k.id == cid) c = &k;
            if (!c) fail("scan names an unknown component");
            c->td = tabs >> 4; c->ta = tabs & 15;
            if (c->td > 3 || c->ta > 3) fail("bad table selector");
            sc.push_back(c);
        }
        const int ss = d[p + 1 + 2 * ns], se = d[p + 2 + 2 * ns], ah = d[p + 3 + 2 * ns] >> 4, al = d[p + 3 + 2 * ns] & 15;
        if (progressive) { if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) fail("bad progressive scan parameters"); }
        else if (ss != 0 || se != 63 || ah != 0 || al != 0) fail("bad sequential scan parameters");
        pos = end;
        reset_entropy();
        uint32_t todo = restart_interval, rst = 0;
        auto one_block = [&](JComp& c, uint32_t bx, uint32_t by) {
            int16_t* blk = &c.coef[((size_t)by * c.bw + bx) * 64];
            if (!progressive) block_sequential(c, blk);
            else if (ss == 0) block_dc_progressive(c, blk, ah, al);
            else block_ac_progressive(c, blk, ss, se, ah, al);
        };
        auto after_unit = [&](bool last) {
            if (restart_interval && !last && --todo == 0) { // the next marker must be RSTn
                bitbuf = 0; bitcnt = 0;
                while (pos + 1 < d.size() && !(d[pos] == 0xff && d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7)) ++pos;
                if (pos + 1 >= d.size() || d[pos + 1] != 0xd0 + (rst & 7)) fail("missing restart marker");
                pos += 2; ++rst; todo = restart_interval;
                reset_entropy();
            }
        };
        if (ns == 1) { // non-interleaved: the component's own blocks that cover the image, row by row
            JComp& c = *sc[0];
            const uint32_t nbx = (((width * (uint32_t)c.h + (uint32_t)hmax - 1) / (uint32_t)hmax) + 7) / 8;
            const uint32_t nby = (((height * (uint32_t)c.v + (uint32_t)vmax - 1) / (uint32_t)vmax) + 7) / 8;
            for (uint32_t by = 0; by < nby; ++by) for (uint32_t bx = 0; bx < nbx; ++bx) { one_block(c, bx, by); after_unit(by + 1 == nby && bx + 1 == nbx); }
        } else {
            for (uint32_t my = 0; my < mcuy; ++my) for (uint32_t mx = 0; mx < mcux; ++mx) {
                for (JComp* c : sc) for (int v = 0; v < c->v; ++v) for (int h = 0; h < c->h; ++h) one_block(*c, mx * (uint32_t)c->h + (uint32_t)h, my * (uint32_t)c->v + (uint32_t)v);
                after_unit(my + 1 == mcuy && mx + 1 == mcux);
            }
        }
        // `pos` now sits somewhere before the next marker (the reader stops at it); the caller scans forward for it
    }

    // ---- inverse DCT: IJG jidctint ("islow") with 12-bit constants, column pass to 10 fractional bits, row pass with the +128 level shift
    static inline int f2f(double x) { return (int)(x * 4096.0 + 0.5); }
    static uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
    static void idct_1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, int& x0, int& x1, int& x2, int& x3, int& t0, int& t1, int& t2, int& t3) {
        int p2 = s2, p3 = s6;
        int p1 = (p2 + p3) * f2f(0.5411961);
        t2 = p1 + p3 * f2f(-1.847759065);
        t3 = p1 + p2 * f2f(0.765366865);
        p2 = s0; p3 = s4;
        t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;
        x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;
        t0 = s7; t1 = s5; t2 = s3; t3 = s1;
        p3 = t0 + t2; int p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
        const int p5 = (p3 + p4) * f2f(1.175875602);
        t0 = t0 * f2f(0.298631336); t1 = t1 * f2f(2.053119869); t2 = t2 * f2f(3.072711026); t3 = t3 * f2f(1.501321110);
        p1 = p5 + p1 * f2f(-0.899976223); p2 = p5 + p2 * f2f(-2.562915447);
        p3 = p3 * f2f(-1.961570560); p4 = p4 * f2f(-0.390180644);
        t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
    }
    static void idct_block(uint8_t* out, size_t stride, const int16_t* in, const uint16_t* q) {
        int val[64];
        for (int i = 0; i < 8; ++i) {
            const int16_t* c = in + i; const uint16_t* qq = q + i; int* v = val + i;
            if (!c[8] && !c[16] && !c[24] && !c[32] && !c[40] && !c[48] && !c[56]) {
                const int dc = c[0] * qq[0] * 4;
                for (int k = 0; k < 8; ++k) v[8 * k] = dc;
                continue;
            }
            int x0, x1, x2, x3, t0, t1, t2, t3;
            idct_1d(c[0] * qq[0], c[8] * qq[8], c[16] * qq[16], c[24] * qq[24], c[32] * qq[32], c[40] * qq[40], c[48] * qq[48], c[56] * qq[56], x0, x1, x2, x3, t0, t1, t2, t3);
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10; v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10; v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
        for (int i = 0; i < 8; ++i) {
            const int* v = val + 8 * i; uint8_t* o = out + stride * (size_t)i;
            int x0, x1, x2, x3, t0, t1, t2, t3;
            idct_1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x0, x1, x2, x3, t0, t1, t2, t3);
            const int bias = 65536 + (128 << 17);
            x0 += bias; x1 += bias; x2 += bias; x3 += bias;
            o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17); o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
            o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17); o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
        }
    }

    // ---- chroma upsampling with the triangle filter (weights 3 : 1 towards the nearer sample along every subsampled axis)
    static void up_h2(const uint8_t* in, uint32_t n, uint8_t* out) { // n samples -> 2n
        if (n == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        for (uint32_t i = 1; i + 1 < n; ++i) { const int t = 3 * in[i] + 2; out[2 * i] = (uint8_t)((t + in[i - 1]) >> 2); out[2 * i + 1] = (uint8_t)((t + in[i + 1]) >> 2); }
        // the second-to-last output weights the sample BEFORE the last one 3 : 1 — stb_image's own edge rule (libjpeg mirrors the
        // left edge instead); kept because the reference's texels come from stb_image
        out[2 * n - 2] = (uint8_t)((in[n - 2] * 3 + in[n - 1] + 2) >> 2); out[2 * n - 1] = in[n - 1];
    }
    static void up_v2(const uint8_t* nearr, const uint8_t* farr, uint32_t n, uint8_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = (uint8_t)((3 * nearr[i] + farr[i] + 2) >> 2); }
    static void up_h2v2(const uint8_t* nearr, const uint8_t* farr, uint32_t n, uint8_t* out) {
        if (n == 1) { out[0] = out[1] = (uint8_t)((3 * nearr[0] + farr[0] + 2) >> 2); return; }
        int t1 = 3 * nearr[0] + farr[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (uint32_t i = 1; i < n; ++i) {
            const int t0 = t1; t1 = 3 * nearr[i] + farr[i];
            out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4); out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[2 * n - 1] = (uint8_t)((t1 + 2) >> 2);
    }

    Image8 finish() {
        for (auto& c : comps) {
            if (!qt_ok[c.tq]) fail("missing quantisation table");
            c.plane.resize((size_t)c.bw * 8 * c.bh * 8);
            for (uint32_t by = 0; by < c.bh; ++by) for (uint32_t bx = 0; bx < c.bw; ++bx)
                idct_block(&c.plane[((size_t)by * 8 * c.bw + bx) * 8], (size_t)c.bw * 8, &c.coef[((size_t)by * c.bw + bx) * 64], qt[c.tq]);
        }
        Image8 out; out.width = width; out.height = height; out.channels = (int)comps.size();
        out.data.resize((size_t)width * height * out.channels);
        std::vector<std::vector<uint8_t>> line(comps.size(), std::vector<uint8_t>((size_t)width + 16 * (size_t)hmax));
        for (uint32_t y = 0; y < height; ++y) {
            for (size_t k = 0; k < comps.size(); ++k) {
                JComp& c = comps[k];
                const uint32_t hs = (uint32_t)(hmax / c.h), vs = (uint32_t)(vmax / c.v);
                const uint32_t cw = (width * (uint32_t)c.h + (uint32_t)hmax - 1) / (uint32_t)hmax, chh = (height * (uint32_t)c.v + (uint32_t)vmax - 1) / (uint32_t)vmax;
                const size_t pstride = (size_t)c.bw * 8;
                const uint32_t cy = y / vs;
                const uint8_t* nearr = &c.plane[pstride * cy];
                uint8_t* o = line[k].data();
                if (hs == 1 && vs == 1) std::memcpy(o, nearr, cw);
                else if (vs == 2 && hs <= 2) {
                    const uint32_t fy = (y & 1u) ? std::min(cy + 1, chh - 1) : (cy ? cy - 1 : 0);
                    const uint8_t* farr = &c.plane[pstride * fy];
                    if (hs == 2) up_h2v2(nearr, farr, cw, o); else up_v2(nearr, farr, cw, o);
                } else if (hs == 2 && vs == 1) up_h2(nearr, cw, o);
                else for (uint32_t x = 0; x < width; ++x) o[x] = nearr[x / hs]; // other ratios: replicate
            }
            uint8_t* o = &out.data[(size_t)y * width * out.channels];
            if (comps.size() == 1) std::memcpy(o, line[0].data(), width);
            else for (uint32_t x = 0; x < width; ++x) { // JFIF YCbCr -> RGB, 20-bit fixed point
                const int yf = (line[0][x] << 20) + (1 << 19), cb = line[1][x] - 128, cr = line[2][x] - 128;
                // 1.402, 0.71414, 0.34414, 1.772 rounded to 12 fractional bits and shifted to 20; the Cb term of G keeps 16 bits
                const int r = yf + cr * 1470208, g = yf - cr * 748800 + (int)((uint32_t)(cb * -360960) & 0xffff0000u), b = yf + cb * 1858048;
                o[3 * x] = clamp8(r >> 20); o[3 * x + 1] = clamp8(g >> 20); o[3 * x + 2] = clamp8(b >> 20);
            }
        }
        return out;
    }

    Image8 run() {
        if (d.size() < 4 || d[0] != 0xff || d[1] != 0xd8) fail("missing SOI");
        pos = 2;
        bool saw_scan = false;
        for (;;) {
            while (pos < d.size() && d[pos] != 0xff) ++pos;       // (after a scan: skip what is left of the entropy-coded segment)
            while (pos < d.size() && d[pos] == 0xff) ++pos;
            if (pos >= d.size()) { if (saw_scan) break; fail("truncated file"); }
            const int m = d[pos++];
            if (m == 0x00 || (m >= 0xd0 && m <= 0xd7)) continue;  // stuffed byte / stray restart marker inside skipped data
            if (m == 0xd9) break;
            if (m == 0xd8 || m == 0x01) continue;
            const size_t len = be16(pos), p = pos + 2, end = pos + len;
            if (len < 2 || end > d.size()) fail("truncated segment");
            switch (m) {
            case 0xdb: parse_dqt(p, end); pos = end; break;
            case 0xc4: parse_dht(p, end); pos = end; break;
            case 0xc0: case 0xc1: parse_sof(p, end, false); pos = end; break;
            case 0xc2: parse_sof(p, end, true); pos = end; break;
            case 0xdd: if (len < 4) fail("bad DRI"); restart_interval = be16(p); pos = end; break;
            case 0xda: scan(p, end); saw_scan = true; break;
            case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                fail("lossless / hierarchical / arithmetic-coded files are not supported");
            default: pos = end; break; // APPn, COM, ...
            }
        }
        if (!width || !saw_scan) fail("no image data");
        return finish();
    }
};

} // namespace

Image8 read_image(const std::string& path) {
    const std::vector<uint8_t> d = slurp(path);
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() >= 8 && !std::memcmp(d.data(), png_sig, 8)) return read_png(path);
    if (d.size() >= 3 && d[0] == 0xff && d[1] == 0xd8 && d[2] == 0xff) { JDec j(d, path); return j.run(); }
    if (d.size() >= 2 && d[0] == 'B' && d[1] == 'M') return read_bmp(d, path);
    return read_tga(d, path); // no signature: recognised by a plausible header, tried last (as stb_image does)
}

} // namespace nrays_host
