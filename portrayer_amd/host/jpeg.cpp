// JPEG reader for image textures (the reference opens textures with the `image` crate, which reads
// JPEG as well as PNG: src/texture.rs:104-141, and 23 of its 25 texture assets are .jpg).
//
// Baseline / extended sequential (SOF0, SOF1) and progressive (SOF2) Huffman JPEG, 8 bits per sample,
// grey or YCbCr, chroma subsampling 4:4:4, 4:2:2 (h2v1) and 4:2:0 (h2v2), restart intervals. The sample
// pipeline follows the IJG library's default decompression path step for step — the "slow" integer
// inverse DCT (LL&M, 13-bit constants, two passes), "fancy" triangle-filter chroma upsampling with
// replicated edges, and the 16-bit fixed-point YCbCr -> RGB tables — so that a decoded texture equals what
// Pillow / libjpeg(-turbo) returns for the same file, bit for bit; tests/test_host_logic.py checks that on
// the committed assets. (The reference's own decoder, the `jpeg-decoder` crate, uses a different integer
// IDCT; its texels differ from these by a level or two on some pixels — decoder noise either way.)
#include <array>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

#include "host_internal.hpp"

namespace portrayer {
namespace detail {
namespace {

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    bool present = false;
    uint8_t bits[17] = {0};
    std::vector<uint8_t> vals;
    int mincode[17], maxcode[18], valptr[17];
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int dc_tbl = 0, ac_tbl = 0;
    size_t blocks_w = 0, blocks_h = 0;   // allocated (whole MCUs)
    size_t width = 0, height = 0;        // downsampled size in samples: ceil(W * h / hmax)
    std::vector<int16_t> coef;           // blocks_h x blocks_w x 64, natural (row-major) order inside a block
    std::vector<uint8_t> plane;          // blocks_h * 8 rows x blocks_w * 8 columns
    int pred = 0;
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    void fill() {
        while (n <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int c = p < end ? *p : 0;
                    if (c == 0) p++;                 // stuffed zero
                    else { hit_marker = true; p--; b = 0; }  // a marker: feed zeros from here on
                }
            }
            acc |= (uint32_t)b << (24 - n);
            n += 8;
        }
    }
    int bit() {
        if (n < 1) fill();
        int b = (int)(acc >> 31);
        acc <<= 1; n--;
        return b;
    }
    int bits(int k) {
        if (k == 0) return 0;
        if (n < k) fill();
        int v = (int)(acc >> (32 - k));
        acc <<= k; n -= k;
        return v;
    }
    void reset() { acc = 0; n = 0; hit_marker = false; }
};

int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }  // F.2.2.1 EXTEND

int decode(BitReader& br, const Huffman& h) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[(size_t)(h.valptr[l] + code - h.mincode[l])];
    }
    throw std::runtime_error("bad Huffman code in JPEG data");
}

// jidctint.c (jpeg_idct_islow): CONST_BITS 13, PASS1_BITS 2
const int32_t FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270, FIX_0_899976223 = 7373,
              FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137, FIX_1_961570560 = 16069, FIX_2_053119869 = 16819,
              FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;
inline int32_t descale(int64_t x, int n) { return (int32_t)((x + ((int64_t)1 << (n - 1))) >> n); }

void idct_islow(const int16_t* coef, const uint16_t* q, uint8_t* out, size_t stride) {
    int32_t ws[64];
    for (int c = 0; c < 8; c++) {  // pass 1: columns
        const int16_t* in = coef + c;
        const uint16_t* qq = q + c;
        int32_t* w = ws + c;
        if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
            int32_t dc = (int32_t)((uint32_t)(in[0] * qq[0]) << 2);
            for (int r = 0; r < 8; r++) w[8 * r] = dc;
            continue;
        }
        int64_t z2 = in[16] * qq[16], z3 = in[48] * qq[48];
        int64_t z1 = (z2 + z3) * FIX_0_541196100;
        int64_t tmp2 = z1 + z3 * -FIX_1_847759065;
        int64_t tmp3 = z1 + z2 * FIX_0_765366865;
        z2 = in[0] * qq[0]; z3 = in[32] * qq[32];
        int64_t tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
        int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56] * qq[56]; tmp1 = in[40] * qq[40]; tmp2 = in[24] * qq[24]; tmp3 = in[8] * qq[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        w[0] = descale(tmp10 + tmp3, 11); w[56] = descale(tmp10 - tmp3, 11);
        w[8] = descale(tmp11 + tmp2, 11); w[48] = descale(tmp11 - tmp2, 11);
        w[16] = descale(tmp12 + tmp1, 11); w[40] = descale(tmp12 - tmp1, 11);
        w[24] = descale(tmp13 + tmp0, 11); w[32] = descale(tmp13 - tmp0, 11);
    }
    for (int r = 0; r < 8; r++) {  // pass 2: rows, + 128, clamp
        const int32_t* w = ws + 8 * r;
        uint8_t* o = out + r * stride;
        auto clamp = [](int32_t v) { v += 128; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * FIX_0_541196100;
        int64_t tmp2 = z1 + z3 * -FIX_1_847759065;
        int64_t tmp3 = z1 + z2 * FIX_0_765366865;
        int64_t tmp0 = ((int64_t)w[0] + w[4]) * 8192, tmp1 = ((int64_t)w[0] - w[4]) * 8192;
        int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        o[0] = clamp(descale(tmp10 + tmp3, 18)); o[7] = clamp(descale(tmp10 - tmp3, 18));
        o[1] = clamp(descale(tmp11 + tmp2, 18)); o[6] = clamp(descale(tmp11 - tmp2, 18));
        o[2] = clamp(descale(tmp12 + tmp1, 18)); o[5] = clamp(descale(tmp12 - tmp1, 18));
        o[3] = clamp(descale(tmp13 + tmp0, 18)); o[4] = clamp(descale(tmp13 - tmp0, 18));
    }
}

struct Decoder {
    std::vector<uint8_t> f;
    size_t pos = 0;
    size_t W = 0, H = 0;
    bool progressive = false;
    int hmax = 1, vmax = 1;
    size_t mcus_x = 0, mcus_y = 0;
    std::vector<Component> comps;
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    int restart_interval = 0;
    int adobe_transform = -1;
    unsigned eobrun = 0;

    uint16_t be16(size_t p) const { return (uint16_t)((f[p] << 8) | f[p + 1]); }

    void parse() {
        if (f.size() < 4 || f[0] != 0xFF || f[1] != 0xD8) throw std::runtime_error("not a JPEG file");
        pos = 2;
        for (;;) {
            while (pos < f.size() && f[pos] != 0xFF) pos++;
            while (pos < f.size() && f[pos] == 0xFF) pos++;
            if (pos >= f.size()) throw std::runtime_error("JPEG ends before EOI");
            int m = f[pos++];
            if (m == 0xD9) break;
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (pos + 2 > f.size()) throw std::runtime_error("truncated JPEG segment");
            size_t len = be16(pos), seg = pos + 2, next = pos + len;
            if (len < 2 || next > f.size()) throw std::runtime_error("truncated JPEG segment");
            switch (m) {
            case 0xC0: case 0xC1: case 0xC2: frame(seg, m == 0xC2); break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                throw std::runtime_error("unsupported JPEG process (lossless / hierarchical / arithmetic coding)");
            case 0xC4: dht(seg, next); break;
            case 0xDB: dqt(seg, next); break;
            case 0xDD: restart_interval = be16(seg); break;
            case 0xEE: if (len >= 14 && std::memcmp(&f[seg], "Adobe", 5) == 0) adobe_transform = f[seg + 11]; break;
            case 0xDA: pos = next; scan(seg); continue;  // scan() leaves pos at the marker that ended the entropy-coded data
            default: break;
            }
            pos = next;
        }
    }

    void frame(size_t p, bool prog) {
        if (!comps.empty()) throw std::runtime_error("JPEG with more than one frame");
        progressive = prog;
        if (f[p] != 8) throw std::runtime_error("only 8-bit JPEG is supported");
        H = be16(p + 1); W = be16(p + 3);
        int n = f[p + 5];
        if (W == 0 || H == 0 || (n != 1 && n != 3)) throw std::runtime_error("unsupported JPEG frame (size 0 or component count not 1 / 3)");
        comps.resize((size_t)n);
        for (int i = 0; i < n; i++) {
            Component& c = comps[(size_t)i];
            c.id = f[p + 6 + 3 * i]; c.h = f[p + 7 + 3 * i] >> 4; c.v = f[p + 7 + 3 * i] & 15; c.tq = f[p + 8 + 3 * i] & 3;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) throw std::runtime_error("bad JPEG sampling factors");
            hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
        }
        mcus_x = (W + 8 * (size_t)hmax - 1) / (8 * (size_t)hmax);
        mcus_y = (H + 8 * (size_t)vmax - 1) / (8 * (size_t)vmax);
        for (Component& c : comps) {
            c.blocks_w = mcus_x * (size_t)c.h; c.blocks_h = mcus_y * (size_t)c.v;
            c.width = (W * (size_t)c.h + (size_t)hmax - 1) / (size_t)hmax;
            c.height = (H * (size_t)c.v + (size_t)vmax - 1) / (size_t)vmax;
            c.coef.assign(c.blocks_w * c.blocks_h * 64, 0);
        }
    }

    void dqt(size_t p, size_t end) {
        while (p < end) {
            int pq = f[p] >> 4, tq = f[p] & 15;
            p++;
            if (tq > 3) throw std::runtime_error("bad JPEG quantisation table id");
            for (int i = 0; i < 64; i++) {
                qt[tq][ZIGZAG[i]] = pq ? be16(p) : f[p];
                p += pq ? 2 : 1;
            }
            qt_present[tq] = true;
        }
    }

    void dht(size_t p, size_t end) {
        while (p < end) {
            int tc = f[p] >> 4, th = f[p] & 15;
            p++;
            if (tc > 1 || th > 3) throw std::runtime_error("bad JPEG Huffman table id");
            Huffman& h = tc ? ac[th] : dc[th];
            int total = 0;
            for (int l = 1; l <= 16; l++) { h.bits[l] = f[p + (size_t)l - 1]; total += h.bits[l]; }
            p += 16;
            h.vals.assign(f.begin() + (long)p, f.begin() + (long)p + total);
            p += (size_t)total;
            h.build();
        }
    }

    // ---- one block of one scan -------------------------------------------------------------------
    void block_baseline(BitReader& br, Component& c, int16_t* b) {
        int t = decode(br, dc[c.dc_tbl]);
        int diff = t ? extend(br.bits(t), t) : 0;
        c.pred += diff;
        b[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            int rs = decode(br, ac[c.ac_tbl]), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) throw std::runtime_error("JPEG coefficient index out of range");
            b[ZIGZAG[k]] = (int16_t)extend(br.bits(s), s);
            k++;
        }
    }
    void block_dc_first(BitReader& br, Component& c, int16_t* b, int al) {
        int t = decode(br, dc[c.dc_tbl]);
        int diff = t ? extend(br.bits(t), t) : 0;
        c.pred += diff;
        b[0] = (int16_t)(c.pred * (1 << al));
    }
    void block_dc_refine(BitReader& br, int16_t* b, int al) {
        if (br.bit()) b[0] = (int16_t)(b[0] | (1 << al));
    }
    void block_ac_first(BitReader& br, Component& c, int16_t* b, int ss, int se, int al) {
        if (eobrun) { eobrun--; return; }
        for (int k = ss; k <= se;) {
            int rs = decode(br, ac[c.ac_tbl]), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1u << r) - 1;
                    if (r) eobrun += (unsigned)br.bits(r);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) throw std::runtime_error("JPEG coefficient index out of range");
            b[ZIGZAG[k]] = (int16_t)(extend(br.bits(s), s) * (1 << al));
            k++;
        }
    }
    void block_ac_refine(BitReader& br, Component& c, int16_t* b, int ss, int se, int al) {  // G.1.2.3 / jdphuff.c decode_mcu_AC_refine
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; k++) {
                int rs = decode(br, ac[c.ac_tbl]), r = rs >> 4, s = rs & 15, value = 0;
                if (s) {
                    value = br.bit() ? p1 : m1;  // s must be 1: a newly non-zero coefficient
                } else if (r != 15) {
                    eobrun = 1u << r;
                    if (r) eobrun += (unsigned)br.bits(r);
                    break;
                }
                // skip r zero-history coefficients (refining the non-zero ones passed on the way), then place the new one
                for (; k <= se; k++) {
                    int16_t* x = b + ZIGZAG[k];
                    if (*x != 0) {
                        if (br.bit() && (*x & p1) == 0) *x = (int16_t)(*x >= 0 ? *x + p1 : *x + m1);
                    } else {
                        if (--r < 0) break;
                    }
                }
                if (value && k <= se) b[ZIGZAG[k]] = (int16_t)value;
            }
        }
        if (eobrun > 0) {  // the rest of the band: only correction bits for coefficients that are already non-zero
            for (; k <= se; k++) {
                int16_t* x = b + ZIGZAG[k];
                if (*x != 0 && br.bit() && (*x & p1) == 0) *x = (int16_t)(*x >= 0 ? *x + p1 : *x + m1);
            }
            eobrun--;
        }
    }

    void scan(size_t p) {
        if (comps.empty()) throw std::runtime_error("JPEG scan before the frame header");
        int ns = f[p++];
        std::vector<Component*> sc;
        for (int i = 0; i < ns; i++) {
            int id = f[p], tbl = f[p + 1];
            p += 2;
            Component* found = nullptr;
            for (Component& c : comps) if (c.id == id) found = &c;
            if (!found) throw std::runtime_error("JPEG scan names an unknown component");
            found->dc_tbl = tbl >> 4; found->ac_tbl = tbl & 15;
            if (found->dc_tbl > 3 || found->ac_tbl > 3) throw std::runtime_error("bad JPEG Huffman table selector");
            sc.push_back(found);
        }
        int ss = f[p], se = f[p + 1], ah = f[p + 2] >> 4, al = f[p + 2] & 15;
        if (!progressive) { ss = 0; se = 63; ah = al = 0; }
        if (ss > se || se > 63 || (ss == 0 && se != 0 && progressive) || (ss > 0 && ns != 1)) throw std::runtime_error("bad JPEG progressive scan parameters");
        for (Component* c : sc) {
            if ((ss == 0 && ah == 0 && !dc[c->dc_tbl].present) || (se > 0 && !ac[c->ac_tbl].present)) throw std::runtime_error("JPEG scan uses a missing Huffman table");
            c->pred = 0;
        }
        eobrun = 0;
        BitReader br;
        br.p = f.data() + pos; br.end = f.data() + f.size();
        auto one_block = [&](Component& c, int16_t* b) {
            if (!progressive) block_baseline(br, c, b);
            else if (ss == 0) { if (ah == 0) block_dc_first(br, c, b, al); else block_dc_refine(br, b, al); }
            else { if (ah == 0) block_ac_first(br, c, b, ss, se, al); else block_ac_refine(br, c, b, ss, se, al); }
        };
        auto restart = [&]() {  // RSTn: byte-align, skip the marker, reset the predictors
            while (br.p < br.end && !(br.p[0] == 0xFF && br.p + 1 < br.end && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) {
                if (br.p[0] == 0xFF && br.p + 1 < br.end && br.p[1] != 0 && br.p[1] != 0xFF) break;  // some other marker: give up looking
                br.p++;
            }
            if (br.p + 1 < br.end && br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7) br.p += 2;
            br.reset();
            for (Component* c : sc) c->pred = 0;
            eobrun = 0;
        };
        size_t done = 0;
        if (ns == 1) {  // non-interleaved: the component's own blocks, row by row
            Component& c = *sc[0];
            size_t bw = (c.width + 7) / 8, bh = (c.height + 7) / 8;
            for (size_t by = 0; by < bh; by++)
                for (size_t bx = 0; bx < bw; bx++) {
                    if (restart_interval && done && done % (size_t)restart_interval == 0) restart();
                    one_block(c, &c.coef[(by * c.blocks_w + bx) * 64]);
                    done++;
                }
        } else {
            for (size_t my = 0; my < mcus_y; my++)
                for (size_t mx = 0; mx < mcus_x; mx++) {
                    if (restart_interval && done && done % (size_t)restart_interval == 0) restart();
                    for (Component* c : sc)
                        for (int v = 0; v < c->v; v++)
                            for (int h = 0; h < c->h; h++)
                                one_block(*c, &c->coef[((my * (size_t)c->v + (size_t)v) * c->blocks_w + mx * (size_t)c->h + (size_t)h) * 64]);
                    done++;
                }
        }
        // continue parsing at the marker that ended the entropy-coded segment
        const uint8_t* q = br.hit_marker ? br.p : br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0 && !(q[1] >= 0xD0 && q[1] <= 0xD7) && q[1] != 0xFF)) q++;
        pos = (size_t)(q - f.data());
    }

    // ---- samples ----------------------------------------------------------------------------------
    void inverse_transform() {
        for (Component& c : comps) {
            if (!qt_present[c.tq]) throw std::runtime_error("JPEG component uses a missing quantisation table");
            size_t stride = c.blocks_w * 8;
            c.plane.assign(stride * c.blocks_h * 8, 0);
            for (size_t by = 0; by < c.blocks_h; by++)
                for (size_t bx = 0; bx < c.blocks_w; bx++)
                    idct_islow(&c.coef[(by * c.blocks_w + bx) * 64], qt[c.tq], &c.plane[by * 8 * stride + bx * 8], stride);
        }
    }

    // jdsample.c: full-size plane of a component (W x H), fancy upsampling for h2v1 / h2v2
    std::vector<uint8_t> full_plane(const Component& c) {
        const size_t stride = c.blocks_w * 8;
        std::vector<uint8_t> out(W * H);
        if (c.h == hmax && c.v == vmax) {
            for (size_t y = 0; y < H; y++) std::memcpy(&out[y * W], &c.plane[y * stride], W);
            return out;
        }
        const bool h2 = hmax == 2 * c.h, v2 = vmax == 2 * c.v, v1 = vmax == c.v;
        if (!h2 || !(v1 || v2)) throw std::runtime_error("unsupported JPEG chroma subsampling (only 4:4:4, 4:2:2 and 4:2:0)");
        const size_t cw = c.width, ch = c.height;
        if (cw <= 2) {  // jdsample.c falls back to plain replication for components this narrow
            for (size_t y = 0; y < H; y++)
                for (size_t x = 0; x < W; x++) out[y * W + x] = c.plane[(v2 ? y / 2 : y) * stride + x / 2];
            return out;
        }
        std::vector<int> colsum(cw);
        std::vector<uint8_t> row(2 * cw);
        for (size_t y = 0; y < H; y++) {
            if (v1) {  // h2v1_fancy_upsample
                const uint8_t* in = &c.plane[y * stride];
                {
                    row[0] = in[0];
                    row[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                    for (size_t x = 1; x + 1 < cw; x++) {
                        int v = in[x] * 3;
                        row[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
                        row[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
                    }
                    int v = in[cw - 1] * 3;
                    row[2 * cw - 2] = (uint8_t)((v + in[cw - 2] + 1) >> 2);
                    row[2 * cw - 1] = in[cw - 1];
                }
            } else {  // h2v2_fancy_upsample: output row y comes from input row y / 2 (weight 3) and its neighbour above (even y) or below (odd y)
                size_t r0 = y / 2;
                size_t r1 = (y & 1) ? (r0 + 1 < ch ? r0 + 1 : ch - 1) : (r0 > 0 ? r0 - 1 : 0);  // edges replicate
                const uint8_t* in0 = &c.plane[r0 * stride];
                const uint8_t* in1 = &c.plane[r1 * stride];
                for (size_t x = 0; x < cw; x++) colsum[x] = in0[x] * 3 + in1[x];
                {
                    row[0] = (uint8_t)((colsum[0] * 4 + 8) >> 4);
                    row[1] = (uint8_t)((colsum[0] * 3 + colsum[1] + 7) >> 4);
                    for (size_t x = 1; x + 1 < cw; x++) {
                        row[2 * x] = (uint8_t)((colsum[x] * 3 + colsum[x - 1] + 8) >> 4);
                        row[2 * x + 1] = (uint8_t)((colsum[x] * 3 + colsum[x + 1] + 7) >> 4);
                    }
                    row[2 * cw - 2] = (uint8_t)((colsum[cw - 1] * 3 + colsum[cw - 2] + 8) >> 4);
                    row[2 * cw - 1] = (uint8_t)((colsum[cw - 1] * 4 + 7) >> 4);
                }
            }
            std::memcpy(&out[y * W], row.data(), W);
        }
        return out;
    }

    void to_rgb(std::vector<uint8_t>* rgb) {
        rgb->resize(W * H * 3);
        std::vector<uint8_t> y = full_plane(comps[0]);
        if (comps.size() == 1) {
            for (size_t i = 0; i < W * H; i++) (*rgb)[3 * i] = (*rgb)[3 * i + 1] = (*rgb)[3 * i + 2] = y[i];
            return;
        }
        std::vector<uint8_t> cb = full_plane(comps[1]), cr = full_plane(comps[2]);
        if (adobe_transform == 0) {  // Adobe marker: the three components are R, G, B
            for (size_t i = 0; i < W * H; i++) { (*rgb)[3 * i] = y[i]; (*rgb)[3 * i + 1] = cb[i]; (*rgb)[3 * i + 2] = cr[i]; }
            return;
        }
        // jdcolor.c build_ycc_rgb_table: 16-bit fixed point
        int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
        for (int i = 0; i < 256; i++) {
            int32_t x = i - 128;
            cr_r[i] = (int32_t)((91881 * (int64_t)x + 32768) >> 16);   // FIX(1.40200)
            cb_b[i] = (int32_t)((116130 * (int64_t)x + 32768) >> 16);  // FIX(1.77200)
            cr_g[i] = -46802 * x;                                      // FIX(0.71414)
            cb_g[i] = -22554 * x + 32768;                              // FIX(0.34414)
        }
        auto clamp = [](int32_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        for (size_t i = 0; i < W * H; i++) {
            int32_t Y = y[i];
            (*rgb)[3 * i] = clamp(Y + cr_r[cr[i]]);
            (*rgb)[3 * i + 1] = clamp(Y + ((cb_g[cb[i]] + cr_g[cr[i]]) >> 16));
            (*rgb)[3 * i + 2] = clamp(Y + cb_b[cb[i]]);
        }
    }
};

}  // namespace

bool jpeg_read(const std::string& path, size_t* width, size_t* height, std::vector<uint8_t>* rgb) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    Decoder d;
    d.f.assign((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    try {
        d.parse();
        if (d.comps.empty()) throw std::runtime_error("no frame");
        d.inverse_transform();
        d.to_rgb(rgb);
    } catch (const std::exception& e) {
        throw std::runtime_error(std::string(e.what()) + ": " + path);
    }
    *width = d.W; *height = d.H;
    return true;
}

}  // namespace detail
}  // namespace portrayer
