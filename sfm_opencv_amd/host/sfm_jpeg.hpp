// sfm_jpeg.hpp -- baseline JPEG decoder for the drivers' imread (host C++, header only).
//
// The reference reads its datasets with cv::imread (NViewReconstuct.cpp:801, TwoViewReconstruct.cpp:116): `.jpg` files, all of
// them baseline sequential 8-bit YCbCr (crazyhorse: 4:2:2, 1024 x 768; desktop / dog: 4:2:0 with restart intervals, 3648 x 2736).
// cv::imread decodes through libjpeg(-turbo) with its defaults [3P]: the accurate integer inverse DCT ("islow", 13-bit constants,
// two passes), "fancy" triangle-filter chroma upsampling for 2:1 horizontal and 2:1 x 2:1 factors, the fixed-point YCbCr -> RGB
// tables, and hands the pixels back as BGR.  This file restates those published algorithms so that the drivers see the pixel values
// OpenCV would see; tests/test_jpeg_cpu.py pins it bit for bit against files decoded by an independent libjpeg build (Pillow) in
// the build container (fixtures + generating script under tests/golden/).
//
// Supported: SOF0 / SOF1 (Huffman, 8-bit), 1 or 3 components, sampling factors 1..2 (other ratios: sample replication), DRI /
// RSTn, 8- and 16-bit quantisation tables.  Not supported (decode_jpeg returns false): progressive (SOF2), arithmetic coding,
// 12-bit, CMYK / Adobe-transform files.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace sfm {
namespace jpeg {

struct Huff {
    uint8_t bits[17] = { 0 }; uint8_t vals[256] = { 0 };
    int mincode[17], maxcode[18], valptr[17];
    uint16_t look[512];                 // 9-bit prefix -> (length << 8) | symbol, 0 = longer code
    bool present = false;
    // false: the code lengths over-subscribe the code space (libjpeg's check in jdhuff.c: at every length the running code must
    // stay below 2^length) -- such a table would index past look[] below and past vals[] while decoding
    bool build()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            if (code > (1 << l) || k > 256) return false;
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memset(look, 0, sizeof look);
        code = 0; k = 0;
        for (int l = 1; l <= 9; ++l) {
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                const int lo = code << (9 - l);
                for (int f = 0; f < (1 << (9 - l)) && lo + f < 512; ++f) look[lo + f] = (uint16_t)((l << 8) | vals[k]);
            }
            code <<= 1;
        }
        return true;
    }
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int n = 0; bool hit_marker = false;
    void fill()
    {
        while (n <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;            // stuffed zero
                    else { hit_marker = true; b = 0; }                    // a marker: feed zeros, leave p on it
                } else ++p;
            }
            acc |= (uint32_t)b << (24 - n);
            n += 8;
        }
    }
    int peek(int k) { if (n < k) fill(); return (int)(acc >> (32 - k)); }
    void skip(int k) { acc <<= k; n -= k; }
    int get(int k) { if (k == 0) return 0; const int v = peek(k); skip(k); return v; }
    void reset() { acc = 0; n = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader& br, const Huff& h)
{
    const int pre = br.peek(9);
    const int e = h.look[pre];
    if (e) { br.skip(e >> 8); return e & 255; }
    int code = br.peek(16);
    for (int l = 10; l <= 16; ++l) {
        const int c = code >> (16 - l);
        if (h.maxcode[l] >= 0 && c <= h.maxcode[l] && c >= h.mincode[l]) { br.skip(l); return h.vals[h.valptr[l] + c - h.mincode[l]]; }
    }
    br.skip(16);
    return 0;                                           // corrupt stream
}
inline int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

static const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

inline uint8_t clamp_sample(int v) { v += 128; return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// jpeg_idct_islow [3P, libjpeg jidctint.c]: Loeffler-Ligtenberg-Moschytz, CONST_BITS = 13, PASS1_BITS = 2; `coef` already dequantised
inline void idct_islow(const int* coef, uint8_t* out, int stride)
{
    const int C0298 = 2446, C0390 = 3196, C0541 = 4433, C0765 = 6270, C0899 = 7373, C1175 = 9633, C1501 = 12299, C1847 = 15137,
              C1961 = 16069, C2053 = 16819, C2562 = 20995, C3072 = 25172;
    int ws[64];
    auto descale = [](long long x, int n) { return (int)((x + (1ll << (n - 1))) >> n); };
    for (int c = 0; c < 8; ++c) {
        const int* in = coef + c;
        long long z2 = in[16], z3 = in[48];
        long long z1 = (z2 + z3) * C0541;
        long long tmp2 = z1 + z3 * (-C1847), tmp3 = z1 + z2 * C0765;
        z2 = in[0]; z3 = in[32];
        long long tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
        const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long long z4 = tmp1 + tmp3;
        const long long z5 = (z3 + z4) * C1175;
        tmp0 *= C0298; tmp1 *= C2053; tmp2 *= C3072; tmp3 *= C1501;
        z1 *= -C0899; z2 *= -C2562; z3 *= -C1961; z4 *= -C0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[c] = descale(tmp10 + tmp3, 11); ws[56 + c] = descale(tmp10 - tmp3, 11);
        ws[8 + c] = descale(tmp11 + tmp2, 11); ws[48 + c] = descale(tmp11 - tmp2, 11);
        ws[16 + c] = descale(tmp12 + tmp1, 11); ws[40 + c] = descale(tmp12 - tmp1, 11);
        ws[24 + c] = descale(tmp13 + tmp0, 11); ws[32 + c] = descale(tmp13 - tmp0, 11);
    }
    for (int r = 0; r < 8; ++r) {
        const int* w = ws + 8 * r;
        long long z2 = w[2], z3 = w[6];
        long long z1 = (z2 + z3) * C0541;
        long long tmp2 = z1 + z3 * (-C1847), tmp3 = z1 + z2 * C0765;
        long long tmp0 = ((long long)w[0] + w[4]) * 8192, tmp1 = ((long long)w[0] - w[4]) * 8192;
        const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long long z4 = tmp1 + tmp3;
        const long long z5 = (z3 + z4) * C1175;
        tmp0 *= C0298; tmp1 *= C2053; tmp2 *= C3072; tmp3 *= C1501;
        z1 *= -C0899; z2 *= -C2562; z3 *= -C1961; z4 *= -C0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        uint8_t* o = out + (size_t)r * stride;
        o[0] = clamp_sample(descale(tmp10 + tmp3, 18)); o[7] = clamp_sample(descale(tmp10 - tmp3, 18));
        o[1] = clamp_sample(descale(tmp11 + tmp2, 18)); o[6] = clamp_sample(descale(tmp11 - tmp2, 18));
        o[2] = clamp_sample(descale(tmp12 + tmp1, 18)); o[5] = clamp_sample(descale(tmp12 - tmp1, 18));
        o[3] = clamp_sample(descale(tmp13 + tmp0, 18)); o[4] = clamp_sample(descale(tmp13 - tmp0, 18));
    }
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int w = 0, hgt = 0;                 // true (downsampled) size: ceil(image * samp / max)
    int pw = 0, ph = 0;                 // plane size padded to whole MCUs
    std::vector<uint8_t> plane;
    int pred = 0;
};

// chroma plane (cw x ch true samples, stride cs) -> full resolution (W x H), libjpeg jdsample.c [3P]
inline void upsample(const Component& c, int hmax, int vmax, int W, int H, std::vector<uint8_t>& out)
{
    out.assign((size_t)W * H, 0);
    const int hr = hmax / c.h, vr = vmax / c.v;
    const uint8_t* src = c.plane.data(); const int cs = c.pw, cw = c.w, ch = c.hgt;
    if (hr == 1 && vr == 1) {
        for (int y = 0; y < H; ++y) memcpy(&out[(size_t)y * W], src + (size_t)y * cs, (size_t)W);
    } else if (hr == 2 && vr == 1 && cw > 2) {          // h2v1_fancy_upsample
        std::vector<uint8_t> row((size_t)2 * cw);
        for (int y = 0; y < H; ++y) {
            const uint8_t* in = src + (size_t)y * cs;
            row[0] = in[0]; row[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
            for (int x = 1; x < cw - 1; ++x) { const int t = in[x] * 3; row[2 * x] = (uint8_t)((t + in[x - 1] + 1) >> 2); row[2 * x + 1] = (uint8_t)((t + in[x + 1] + 2) >> 2); }
            row[2 * cw - 2] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2); row[2 * cw - 1] = in[cw - 1];
            memcpy(&out[(size_t)y * W], row.data(), (size_t)W);
        }
    } else if (hr == 2 && vr == 2 && cw > 2) {          // h2v2_fancy_upsample: 3/4 nearer row + 1/4 further row, then 3/4 + 1/4 across
        std::vector<uint8_t> row((size_t)2 * cw);
        for (int y = 0; y < H; ++y) {
            const int iy = y >> 1;
            int ny = (y & 1) ? iy + 1 : iy - 1;
            if (ny < 0) ny = 0;
            if (ny > ch - 1) ny = ch - 1;
            const uint8_t* in0 = src + (size_t)iy * cs; const uint8_t* in1 = src + (size_t)ny * cs;
            int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
            row[0] = (uint8_t)((thiscol * 4 + 8) >> 4); row[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol; thiscol = nextcol;
            for (int x = 1; x < cw - 1; ++x) {
                nextcol = in0[x + 1] * 3 + in1[x + 1];
                row[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); row[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol; thiscol = nextcol;
            }
            row[2 * cw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); row[2 * cw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
            memcpy(&out[(size_t)y * W], row.data(), (size_t)W);
        }
    } else {                                            // int_upsample: replication
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) out[(size_t)y * W + x] = src[(size_t)(y / vr) * cs + x / hr];
    }
}

// out: rows x cols x channels (3 = BGR as cv::imread returns it, 1 = gray); false on anything unsupported or corrupt
inline bool decode_jpeg(const uint8_t* d, size_t n, int& rows, int& cols, int& channels, std::vector<uint8_t>& out)
{
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
    uint16_t qt[4][64]; bool qt_ok[4] = { false, false, false, false };
    Huff hdc[4], hac[4];
    std::vector<Component> comp;
    int W = 0, H = 0, restart = 0, adobe_transform = -1;
    size_t i = 2;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) return false;
        while (i < n && d[i] == 0xFF) ++i;              // fill bytes
        if (i >= n) return false;
        const int m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return false;                    // EOI before any scan
        if (i + 2 > n) return false;
        const size_t L = ((size_t)d[i] << 8) | d[i + 1];
        if (L < 2 || i + L > n) return false;
        const uint8_t* s = d + i + 2; const size_t sl = L - 2;
        if (m == 0xDB) {                                // DQT
            size_t k = 0;
            while (k < sl) {
                const int pq = s[k] >> 4, tq = s[k] & 15; ++k;
                if (tq > 3 || k + (pq ? 128 : 64) > sl) return false;
                for (int z = 0; z < 64; ++z) { qt[tq][kZigzag[z]] = pq ? (uint16_t)((s[k] << 8) | s[k + 1]) : s[k]; k += pq ? 2 : 1; }
                qt_ok[tq] = true;
            }
        } else if (m == 0xC4) {                         // DHT
            size_t k = 0;
            while (k + 17 <= sl) {
                const int tc = s[k] >> 4, th = s[k] & 15; ++k;
                if (tc > 1 || th > 3) return false;
                Huff& h = tc ? hac[th] : hdc[th];
                int total = 0;
                h.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = s[k++]; total += h.bits[l]; }
                if (total > 256 || k + total > sl) return false;
                memcpy(h.vals, s + k, (size_t)total); k += total;
                if (!h.build()) return false;
                h.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {            // SOF0 / SOF1
            if (sl < 6 || s[0] != 8) return false;
            H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4];
            const int nc = s[5];
            if ((nc != 1 && nc != 3) || sl < (size_t)(6 + 3 * nc) || W <= 0 || H <= 0 || (size_t)W * H > ((size_t)1 << 28)) return false;
            comp.resize(nc);
            for (int c = 0; c < nc; ++c) { comp[c].id = s[6 + 3 * c]; comp[c].h = s[7 + 3 * c] >> 4; comp[c].v = s[7 + 3 * c] & 15; comp[c].tq = s[8 + 3 * c]; }
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return false;                               // progressive / lossless / arithmetic
        } else if (m == 0xDD) {
            if (sl < 2) return false;
            restart = (s[0] << 8) | s[1];
        } else if (m == 0xEE) {                         // Adobe APP14
            if (sl >= 12 && memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11];
        } else if (m == 0xDA) {                         // SOS: the one scan of a baseline file
            if (comp.empty() || sl < 1) return false;
            const int ns = s[0];
            if (ns != (int)comp.size() || sl < (size_t)(1 + 2 * ns + 3)) return false;
            for (int k = 0; k < ns; ++k) {
                const int cid = s[1 + 2 * k];
                bool found = false;
                for (auto& c : comp) if (c.id == cid) { c.td = s[2 + 2 * k] >> 4; c.ta = s[2 + 2 * k] & 15; found = true; }
                if (!found) return false;
            }
            i += L;
            break;
        }
        i += L;
    }
    if (comp.empty() || i >= n) return false;
    if (comp.size() == 3 && adobe_transform == 0) return false;         // RGB-coded file
    int hmax = 1, vmax = 1;
    for (auto& c : comp) {
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3 || !qt_ok[c.tq] || c.td > 3 || c.ta > 3 || !hdc[c.td].present || !hac[c.ta].present) return false;
        hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax;
    }
    if (comp.size() == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }       // a single-component scan is not interleaved
    for (auto& c : comp) if (hmax % c.h || vmax % c.v) return false;
    const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
    for (auto& c : comp) {
        c.w = (W * c.h + hmax - 1) / hmax; c.hgt = (H * c.v + vmax - 1) / vmax;
        c.pw = mcux * 8 * c.h; c.ph = mcuy * 8 * c.v;
        c.plane.assign((size_t)c.pw * c.ph, 0);
    }
    BitReader br; br.p = d + i; br.end = d + n;
    int coef[64];
    int until_restart = restart;
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            if (restart && until_restart == 0) {
                // byte-align, expect RSTn
                br.reset();
                const uint8_t* p = br.p;
                while (p + 1 < br.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
                if (p + 1 >= br.end) return false;
                br.p = p + 2;
                for (auto& c : comp) c.pred = 0;
                until_restart = restart;
            }
            for (auto& c : comp)
                for (int by = 0; by < c.v; ++by)
                    for (int bx = 0; bx < c.h; ++bx) {
                        memset(coef, 0, sizeof coef);
                        const uint16_t* q = qt[c.tq];
                        int s = decode_symbol(br, hdc[c.td]);
                        if (s > 11) return false;                 // DC difference category of 8-bit data: 0..11 (ITU T.81 F.1.2.1)
                        c.pred += extend(br.get(s), s);
                        if (c.pred < -32768 || c.pred > 32767) return false;      // outside what any encoder of 8-bit samples emits
                        coef[0] = c.pred * q[0];
                        for (int k = 1; k < 64;) {
                            const int rs = decode_symbol(br, hac[c.ta]);
                            const int r = rs >> 4; s = rs & 15;
                            if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                            k += r;
                            if (k > 63 || s > 10) return false;       // AC category of 8-bit data: 1..10
                            const int z = kZigzag[k];
                            coef[z] = extend(br.get(s), s) * q[z];
                            ++k;
                        }
                        idct_islow(coef, &c.plane[(size_t)((my * c.v + by) * 8) * c.pw + (mx * c.h + bx) * 8], c.pw);
                    }
            if (restart) --until_restart;
        }
    rows = H; cols = W;
    if (comp.size() == 1) {
        channels = 1;
        out.resize((size_t)W * H);
        for (int y = 0; y < H; ++y) memcpy(&out[(size_t)y * W], &comp[0].plane[(size_t)y * comp[0].pw], (size_t)W);
        return true;
    }
    channels = 3;
    std::vector<uint8_t> full[3];
    for (int c = 0; c < 3; ++c) upsample(comp[c], hmax, vmax, W, H, full[c]);
    // ycc_rgb_convert [3P, libjpeg jdcolor.c]: 16-bit fixed-point tables
    int cr_r[256], cb_b[256]; long cr_g[256], cb_g[256];
    auto fix = [](double x) { return (long)(x * 65536.0 + 0.5); };
    for (int k = 0; k < 256; ++k) {
        const long x = k - 128;
        cr_r[k] = (int)((fix(1.40200) * x + 32768) >> 16);
        cb_b[k] = (int)((fix(1.77200) * x + 32768) >> 16);
        cr_g[k] = -fix(0.71414) * x;
        cb_g[k] = -fix(0.34414) * x + 32768;
    }
    auto lim = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
    out.resize((size_t)W * H * 3);
    for (size_t p = 0; p < (size_t)W * H; ++p) {
        const int y = full[0][p], cb = full[1][p], cr = full[2][p];
        out[3 * p + 2] = lim(y + cr_r[cr]);
        out[3 * p + 1] = lim(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
        out[3 * p + 0] = lim(y + cb_b[cb]);
    }
    return true;
}

}  // namespace jpeg
}  // namespace sfm
