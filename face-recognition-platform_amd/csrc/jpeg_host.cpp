// Host half of the JPEG ingest: header parsing and Huffman (entropy) decoding of BASELINE JPEG stills into
// quantised DCT coefficients.  The device half (jpeg_kernels.hip) dequantises, runs the inverse DCT, upsamples the chroma
// planes and converts to BGR straight into the engine's staging frame buffer.
//
// Replaces, for uploaded stills: `face_recognition.load_image_file` / PIL decode behind backend/app/routes/face.py:177-185,216
// and backend/app/services/face_service.py:139 - the part of it that is inherently serial (the bit stream); everything
// with data parallelism moves to the GPU.
//
// Written from the JPEG specification (ITU-T T.81): markers SOI / APPn / DQT / SOF0 / SOF1 / DHT / DRI / SOS / RSTn / EOI,
// 8-bit samples, Huffman coding, interleaved or single-component scans that cover all coefficients (Ss = 0, Se = 63); 1 or 3
// components.  Progressive / arithmetic / 12-bit / multi-scan files are reported as unsupported (FRP_ERR_INVALID), never
// half-decoded.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "frp.h"
#include "jpeg_host.h"

namespace frp {

namespace {

const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int kFast = 10;                     // look-ahead bits of the two direct tables

struct HuffTable {
    bool present = false;
    // canonical decoding (T.81 F.2.2.3): per code length the smallest code, the largest code and the value index of the first
    int32_t mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    // look-ahead on the next kFast bits: (length << 8) | value, 0 = code longer than kFast bits
    uint16_t fast[1 << kFast];
    // AC tables only - code AND magnitude bits inside the look-ahead: (coefficient << 8) | (run << 4) | (code + magnitude
    // length); 0 = take the general path.  (Most AC symbols of a photographic still are short codes with 1-3 magnitude bits.)
    int16_t fast_ac[1 << kFast];
};

bool build_table(HuffTable& t, const uint8_t* counts, const uint8_t* vals, int nvals) {
    int code = 0, k = 0;
    memset(t.fast, 0, sizeof(t.fast));
    memset(t.fast_ac, 0, sizeof(t.fast_ac));
    for (int len = 1; len <= 16; ++len) {
        t.valptr[len] = k;
        t.mincode[len] = code;
        for (int i = 0; i < counts[len - 1]; ++i, ++k) {
            if (k >= nvals) return false;
            if (len <= kFast) {
                const int first = code << (kFast - len);
                for (int f = 0; f < (1 << (kFast - len)); ++f) t.fast[first + f] = (uint16_t)((len << 8) | vals[k]);
            }
            ++code;
        }
        t.maxcode[len] = counts[len - 1] ? code - 1 : -1;
        if (code > (1 << len)) return false;           // over-subscribed
        code <<= 1;
    }
    t.maxcode[17] = 0x7fffffff;
    memcpy(t.vals, vals, (size_t)nvals);
    for (int i = 0; i < (1 << kFast); ++i) {
        const uint16_t f = t.fast[i];
        if (!f) continue;
        const int len = f >> 8, run = (f >> 4) & 15, mag = f & 15;
        if (mag == 0 || len + mag > kFast) continue;
        int v = (i >> (kFast - len - mag)) & ((1 << mag) - 1);          // the magnitude bits behind the code
        if (v < (1 << (mag - 1))) v += (int)((~0u) << mag) + 1;          // T.81 F.2.2.1 EXTEND
        if (v >= -128 && v <= 127) t.fast_ac[i] = (int16_t)(v * 256 + run * 16 + len + mag);
    }
    t.present = true;
    return true;
}

// The "typical" Huffman tables of T.81 Annex K.3 (tables K.3 - K.6).  Motion-JPEG sources (AVI "MJPG" chunks, many USB and
// IP cameras) leave the DHT segment out when their frames use exactly these; a decoder installs them before the headers are
// read and a DHT segment, when present, replaces them (what libjpeg-turbo - the decoder behind PIL - does as well).
const uint8_t kStdDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kStdDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kStdDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kStdAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
const uint8_t kStdAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91,
    0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53,
    0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79,
    0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
    0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kStdAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
const uint8_t kStdAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14,
    0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17,
    0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a,
    0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78,
    0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
    0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

// 64-bit window on the entropy-coded segment, left-aligned: the next bit of the stream is bit 63.  Refilled eight bytes at
// a time while none of them is 0xFF (no stuffing, no marker), byte by byte around those.
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;      // a marker (other than stuffing) was reached: only zero bits are fed from here on
    int pad_bits = 0;             // zero bits fed behind a marker / the end of the data (still in the window, or consumed)
    // true when the decoder has consumed bits that were not in the file (the window holds fewer bits than were padded in)
    bool ran_dry() const { return nbits < pad_bits; }
    void fill_slow() {
        while (nbits <= 56) {
            uint32_t b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;        // stuffed zero
                    else { hit_marker = true; b = 0; }
                } else {
                    ++p;
                }
            } else {
                hit_marker = true;
            }
            if (hit_marker) pad_bits += 8;
            acc |= (uint64_t)b << (56 - nbits);
            nbits += 8;
        }
    }
    inline void fill() {
        if (end - p >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            const uint64_t inv = ~w;
            if (!((inv - 0x0101010101010101ull) & ~inv & 0x8080808080808080ull)) {          // no 0xFF among the eight
                w = __builtin_bswap64(w);
                const int k = (64 - nbits) >> 3;                                               // whole bytes that fit
                if (k == 8) acc = w;                                                           // (nbits == 0: the shifts below would be by 64)
                else acc |= (w >> nbits) & (~0ull << (64 - nbits - 8 * k));
                p += k;
                nbits += 8 * k;
                return;
            }
        }
        fill_slow();
    }
    inline uint32_t peek(int n) { return (uint32_t)(acc >> (64 - n)); }
    inline void skip(int n) { acc <<= n; nbits -= n; }
    // callers keep at least 32 bits in the window (a code is at most 16 bits, a magnitude at most 15)
    inline int32_t receive_extend(int s) {
        if (s == 0) return 0;
        const int32_t v = (int32_t)peek(s);
        skip(s);
        return v < (1 << (s - 1)) ? v + (int32_t)((~0u) << s) + 1 : v;
    }
    inline int decode(const HuffTable& t) {
        const uint16_t f = t.fast[peek(kFast)];
        if (f) { skip(f >> 8); return f & 0xff; }
        for (int len = kFast + 1; len <= 16; ++len) {
            const int32_t code = (int32_t)peek(len);
            if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len]) {
                skip(len);
                return t.vals[t.valptr[len] + code - t.mincode[len]];
            }
        }
        return -1;
    }
    void reset_at(const uint8_t* q) { p = q; acc = 0; nbits = 0; hit_marker = false; pad_bits = 0; }
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

}  // namespace

struct JpegHeaderInternal {
    frp_jpeg_info info{};
    uint16_t qt[4][64];          // natural order
    bool have_qt[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    int comp_id[3], comp_tq[3], comp_td[3], comp_ta[3];
    const uint8_t* scan = nullptr;     // first entropy-coded byte
    std::string err;
};

// Largest image (width x height) the decoder accepts: the SOF dimensions come from untrusted bytes and size every buffer
// behind them (a 200-byte file can claim 65,535 x 65,535).  Default = PIL's Image.MAX_IMAGE_PIXELS (the reference decodes
// uploads with PIL, face_service.py:139); FRP_JPEG_MAX_PIXELS overrides it (read once).
long long jpeg_max_pixels() {
    static const long long lim = [] {
        const char* e = getenv("FRP_JPEG_MAX_PIXELS");
        const long long v = e ? atoll(e) : 0;
        return v > 0 ? v : 89478485LL;
    }();
    return lim;
}

static int parse_headers(const uint8_t* d, size_t n, JpegHeaderInternal& H) {
    if (!d || n < 4 || d[0] != 0xFF || d[1] != 0xD8) { H.err = "not a JPEG (no SOI)"; return FRP_ERR_INVALID; }
    build_table(H.dc[0], kStdDcLumaBits, kStdDcVals, 12);                  // Annex K defaults; a DHT segment replaces them
    build_table(H.dc[1], kStdDcChromaBits, kStdDcVals, 12);
    build_table(H.ac[0], kStdAcLumaBits, kStdAcLumaVals, 162);
    build_table(H.ac[1], kStdAcChromaBits, kStdAcChromaVals, 162);
    size_t pos = 2;
    bool have_sof = false, saw_jfif = false, saw_adobe = false;
    int adobe_transform = 0;
    frp_jpeg_info& I = H.info;
    while (pos + 4 <= n) {
        if (d[pos] != 0xFF) { H.err = "marker expected"; return FRP_ERR_INVALID; }
        while (pos < n && d[pos] == 0xFF) ++pos;                     // fill bytes
        if (pos >= n) break;
        const int m = d[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > n) break;
        const int len = be16(d + pos);
        if (len < 2 || pos + (size_t)len > n) { H.err = "truncated segment"; return FRP_ERR_INVALID; }
        const uint8_t* s = d + pos + 2;
        const int sl = len - 2;
        if (m == 0xDB) {                                              // DQT
            int o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                ++o;
                if (tq > 3 || pq > 1 || o + (pq ? 128 : 64) > sl) { H.err = "bad DQT"; return FRP_ERR_INVALID; }
                for (int i = 0; i < 64; ++i) {
                    const int v = pq ? be16(s + o + 2 * i) : s[o + i];
                    H.qt[tq][kZigZag[i]] = (uint16_t)v;
                }
                o += pq ? 128 : 64;
                H.have_qt[tq] = true;
            }
        } else if (m == 0xC4) {                                       // DHT
            int o = 0;
            while (o + 17 <= sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                int total = 0;
                for (int i = 0; i < 16; ++i) total += s[o + 1 + i];
                if (tc > 1 || th > 3 || total > 256 || o + 17 + total > sl) { H.err = "bad DHT"; return FRP_ERR_INVALID; }
                if (!build_table(tc ? H.ac[th] : H.dc[th], s + o + 1, s + o + 17, total)) { H.err = "bad Huffman table"; return FRP_ERR_INVALID; }
                o += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {                          // SOF0 / SOF1: sequential Huffman
            if (sl < 6 || s[0] != 8) { H.err = "only 8-bit samples are supported"; return FRP_ERR_INVALID; }
            I.height = be16(s + 1);
            I.width = be16(s + 3);
            I.components = s[5];
            if (!(I.components == 1 || I.components == 3) || sl < 6 + 3 * I.components || I.width <= 0 || I.height <= 0) {
                H.err = "unsupported component count or size";
                return FRP_ERR_INVALID;
            }
            int hmax = 1, vmax = 1;
            for (int c = 0; c < I.components; ++c) {
                H.comp_id[c] = s[6 + 3 * c];
                I.h_samp[c] = s[7 + 3 * c] >> 4;
                I.v_samp[c] = s[7 + 3 * c] & 15;
                H.comp_tq[c] = s[8 + 3 * c];
                if (I.h_samp[c] < 1 || I.h_samp[c] > 2 || I.v_samp[c] < 1 || I.v_samp[c] > 2 || H.comp_tq[c] > 3) {
                    H.err = "unsupported sampling factors";
                    return FRP_ERR_INVALID;
                }
                hmax = I.h_samp[c] > hmax ? I.h_samp[c] : hmax;
                vmax = I.v_samp[c] > vmax ? I.v_samp[c] : vmax;
            }
            if (I.components == 3 && (I.h_samp[1] != 1 || I.v_samp[1] != 1 || I.h_samp[2] != 1 || I.v_samp[2] != 1 ||
                                      (I.h_samp[0] == 1 && I.v_samp[0] == 2))) {
                H.err = "unsupported chroma subsampling (4:4:4, 4:2:2 and 4:2:0 files are covered)";
                return FRP_ERR_INVALID;
            }
            if (I.components == 1) { I.h_samp[0] = I.v_samp[0] = 1; hmax = vmax = 1; }     // a single component is never interleaved
            if ((long long)I.width * I.height > jpeg_max_pixels()) { H.err = "image exceeds the pixel limit (FRP_JPEG_MAX_PIXELS)"; return FRP_ERR_INVALID; }
            I.mcus_x = (I.width + 8 * hmax - 1) / (8 * hmax);
            I.mcus_y = (I.height + 8 * vmax - 1) / (8 * vmax);
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            H.err = "progressive / lossless / arithmetic-coded JPEG is not supported (baseline only)";
            I.progressive = 1;
            return FRP_ERR_INVALID;
        } else if (m == 0xE0) {                                       // APP0: JFIF fixes the colour space (YCbCr / grey)
            if (sl >= 5 && memcmp(s, "JFIF\0", 5) == 0) saw_jfif = true;
        } else if (m == 0xEE) {                                       // APP14: Adobe transform flag (0 = RGB / CMYK as stored, 1 = YCbCr)
            if (sl >= 12 && memcmp(s, "Adobe", 5) == 0) { saw_adobe = true; adobe_transform = s[11]; }
        } else if (m == 0xDD) {                                       // DRI
            if (sl < 2) { H.err = "bad DRI"; return FRP_ERR_INVALID; }
            I.restart_interval = be16(s);
        } else if (m == 0xDA) {                                       // SOS
            if (!have_sof) { H.err = "SOS before SOF"; return FRP_ERR_INVALID; }
            if (sl < 1) { H.err = "bad SOS"; return FRP_ERR_INVALID; }
            const int ns = s[0];
            if (ns != I.components || sl < 1 + 2 * ns + 3) { H.err = "multi-scan files are not supported"; return FRP_ERR_INVALID; }
            for (int c = 0; c < ns; ++c) {
                if (s[1 + 2 * c] != H.comp_id[c]) { H.err = "scan component order differs from the frame's"; return FRP_ERR_INVALID; }
                H.comp_td[c] = s[2 + 2 * c] >> 4;
                H.comp_ta[c] = s[2 + 2 * c] & 15;
                if (H.comp_td[c] > 3 || H.comp_ta[c] > 3 || !H.dc[H.comp_td[c]].present || !H.ac[H.comp_ta[c]].present || !H.have_qt[H.comp_tq[c]]) {
                    H.err = "scan refers to a table the file does not define";
                    return FRP_ERR_INVALID;
                }
            }
            // colour space of a 3-component file as libjpeg decides it (jdapimin.c: default_decompress_parms), which PIL - the
            // reference's decoder - follows: JFIF -> YCbCr; else Adobe transform 0 -> RGB; else component ids 'R','G','B' -> RGB.
            // The device path converts YCbCr only: RGB-stored files are reported unsupported and take the host decoder.
            if (I.components == 3 && !saw_jfif) {
                const bool rgb_ids = H.comp_id[0] == 'R' && H.comp_id[1] == 'G' && H.comp_id[2] == 'B';
                if (saw_adobe ? adobe_transform == 0 : rgb_ids) { H.err = "RGB-stored JPEG (Adobe transform 0 / RGB component ids) is not supported"; return FRP_ERR_INVALID; }
            }
            if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63) { H.err = "spectral selection is not supported (baseline only)"; return FRP_ERR_INVALID; }
            H.scan = d + pos + len;
            return FRP_OK;
        }
        pos += (size_t)len;
    }
    H.err = "no scan found";
    return FRP_ERR_INVALID;
}

size_t jpeg_coef_elems(const frp_jpeg_info& I) {
    size_t e = 0;
    for (int c = 0; c < I.components; ++c) e += (size_t)I.mcus_x * I.h_samp[c] * I.mcus_y * I.v_samp[c] * 64;
    return e;
}

int jpeg_info(const uint8_t* data, size_t size, frp_jpeg_info* out, std::string* err) {
    JpegHeaderInternal H;
    const int rc = parse_headers(data, size, H);
    if (out) *out = H.info;
    if (rc != FRP_OK && err) *err = H.err;
    return rc;
}

// Entropy-decode every block.  coef: per component a [blocks_y][blocks_x][64] int16 array (natural order, NOT dequantised),
// components back to back; qtab_out: [3][64] uint16 quantisation steps in natural order.
int jpeg_decode_coefficients(const uint8_t* data, size_t size, int16_t* coef, size_t coef_elems, uint16_t* qtab_out, frp_jpeg_info* info_out,
                             std::string* err) {
    JpegHeaderInternal H;
    int rc = parse_headers(data, size, H);
    if (info_out) *info_out = H.info;
    if (rc != FRP_OK) { if (err) *err = H.err; return rc; }
    const frp_jpeg_info& I = H.info;
    if (!coef || coef_elems < jpeg_coef_elems(I)) { if (err) *err = "coefficient buffer too small"; return FRP_ERR_INVALID; }
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 64; ++i) qtab_out[c * 64 + i] = c < I.components ? H.qt[H.comp_tq[c]][i] : 1;
    int16_t* base[3];
    int bx[3], by[3];
    size_t off = 0;
    for (int c = 0; c < I.components; ++c) {
        bx[c] = I.mcus_x * I.h_samp[c];
        by[c] = I.mcus_y * I.v_samp[c];
        base[c] = coef + off;
        off += (size_t)bx[c] * by[c] * 64;
    }
    BitReader br;
    br.reset_at(H.scan);
    br.end = data + size;
    int pred[3] = {0, 0, 0};
    int restart_left = I.restart_interval, next_rst = 0;
    for (int my = 0; my < I.mcus_y; ++my)
        for (int mx = 0; mx < I.mcus_x; ++mx) {
            if (I.restart_interval && restart_left == 0) {
                // byte-align, expect RSTn
                if (br.ran_dry()) { if (err) *err = "entropy-coded data ends before the restart interval is complete"; return FRP_ERR_INVALID; }
                const uint8_t* q = br.p;
                while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;       // (br.p sits at or before the marker)
                if (q + 1 >= br.end || q[1] != 0xD0 + next_rst) { if (err) *err = "restart marker missing or out of sequence"; return FRP_ERR_INVALID; }
                br.reset_at(q + 2);
                next_rst = (next_rst + 1) & 7;
                pred[0] = pred[1] = pred[2] = 0;
                restart_left = I.restart_interval;
            }
            for (int c = 0; c < I.components; ++c) {
                const HuffTable& dct = H.dc[H.comp_td[c]];
                const HuffTable& act = H.ac[H.comp_ta[c]];
                for (int v = 0; v < I.v_samp[c]; ++v)
                    for (int hh = 0; hh < I.h_samp[c]; ++hh) {
                        int16_t* blk = base[c] + ((size_t)(my * I.v_samp[c] + v) * bx[c] + (mx * I.h_samp[c] + hh)) * 64;
                        memset(blk, 0, 64 * sizeof(int16_t));          // here, not over the whole buffer up front: the block is about to be written anyway
                        if (br.nbits < 32) br.fill();
                        const int s = br.decode(dct);
                        if (s < 0 || s > 11) { if (err) *err = "corrupt DC code"; return FRP_ERR_INVALID; }
                        pred[c] += br.receive_extend(s);
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            if (br.nbits < 32) br.fill();
                            const int fa = act.fast_ac[br.peek(kFast)];
                            if (fa) {                                  // code + magnitude in one look-up
                                k += (fa >> 4) & 15;
                                if (k > 63) { if (err) *err = "corrupt AC run"; return FRP_ERR_INVALID; }
                                br.skip(fa & 15);
                                blk[kZigZag[k++]] = (int16_t)(fa >> 8);
                                continue;
                            }
                            const int rs = br.decode(act);
                            if (rs < 0) { if (err) *err = "corrupt AC code"; return FRP_ERR_INVALID; }
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (r == 15) { k += 16; continue; }
                                break;                                 // end of block
                            }
                            k += r;
                            if (k > 63) { if (err) *err = "corrupt AC run"; return FRP_ERR_INVALID; }
                            blk[kZigZag[k]] = (int16_t)br.receive_extend(sz);
                            ++k;
                        }
                    }
            }
            if (I.restart_interval) --restart_left;
        }
    // Bits that were not in the file were consumed: the scan ends (end of data, or a marker) before its last MCU.  libjpeg pads
    // such a scan with a warning and PIL raises "image file is truncated"; here it is an error, so the caller's host path
    // (PIL) decides - never a silently grey frame.
    if (br.ran_dry()) { if (err) *err = "entropy-coded data ends before the last MCU (truncated file)"; return FRP_ERR_INVALID; }
    return FRP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Plan of a device-side entropy decode (restart-interval streams).  The host's part shrinks to the headers and ONE pass over
// the scan for 0xFF bytes: every RSTn marker starts an independent interval (T.81 F.1.1.5: the DC predictors and the bit
// alignment reset there).
static void flatten_table(const HuffTable& t, JpegHuffTableDev& d) {
    memset(&d, 0, sizeof(d));
    for (int i = 0; i < 512; ++i) {
        const uint16_t f = t.fast[i << (kFast - 9)];
        d.fast[i] = (f && (f >> 8) <= 9) ? f : 0;
    }
    for (int l = 0; l < 17; ++l) { d.mincode[l] = t.mincode[l]; d.valptr[l] = t.valptr[l]; }
    for (int l = 0; l < 18; ++l) d.maxcode[l] = t.maxcode[l];
    d.mincode[0] = d.valptr[0] = 0;
    d.maxcode[0] = -1;
    memcpy(d.vals, t.vals, 256);
}

int jpeg_plan_device_decode(const uint8_t* data, size_t size, JpegDevicePlan& plan, JpegHuffTableDev* tables6, std::string* err) {
    JpegHeaderInternal H;
    const int rc = parse_headers(data, size, H);
    plan.info = H.info;
    if (rc != FRP_OK) { if (err) *err = H.err; return rc; }
    const frp_jpeg_info& I = H.info;
    if (I.restart_interval <= 0) { if (err) *err = "no restart intervals"; return FRP_ERR_INVALID; }
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 64; ++i) plan.qtab[c * 64 + i] = c < I.components ? H.qt[H.comp_tq[c]][i] : 1;
    for (int c = 0; c < 3; ++c) {
        const int cc = c < I.components ? c : 0;
        flatten_table(H.dc[H.comp_td[cc]], tables6[2 * c]);
        flatten_table(H.ac[H.comp_ta[cc]], tables6[2 * c + 1]);
    }
    const long mcus = (long)I.mcus_x * I.mcus_y;
    const long n_int = (mcus + I.restart_interval - 1) / I.restart_interval;
    if (n_int > 0x7fffff) { if (err) *err = "too many restart intervals"; return FRP_ERR_INVALID; }
    plan.scan = H.scan;
    plan.int_off.clear();
    plan.int_off.reserve((size_t)n_int + 1);
    plan.int_off.push_back(0);
    const uint8_t* p = H.scan;
    const uint8_t* const end = data + size;
    int next_rst = 0;
    const uint8_t* stop = end;                       // the first marker that is not RSTn (EOI normally), or the end of the file
    while (p < end) {
        p = (const uint8_t*)memchr(p, 0xFF, (size_t)(end - p));
        if (!p || p + 1 >= end) break;
        const int m = p[1];
        if (m == 0x00) { p += 2; continue; }         // stuffed byte
        if (m == 0xFF) { ++p; continue; }            // fill byte
        if (m >= 0xD0 && m <= 0xD7) {
            if (m != 0xD0 + next_rst || (long)plan.int_off.size() >= n_int) { if (err) *err = "restart marker missing or out of sequence"; return FRP_ERR_INVALID; }
            next_rst = (next_rst + 1) & 7;
            plan.int_off.push_back((uint32_t)(p + 2 - H.scan));
            p += 2;
            continue;
        }
        stop = p;                                    // EOI / another segment: the scan ends here
        break;
    }
    if ((long)plan.int_off.size() != n_int) { if (err) *err = "restart marker missing or out of sequence"; return FRP_ERR_INVALID; }
    if ((size_t)(stop - H.scan) >= 0xfffffff0u) { if (err) *err = "scan too large"; return FRP_ERR_INVALID; }
    plan.int_off.push_back((uint32_t)(stop - H.scan));
    plan.scan_bytes = (size_t)(stop - H.scan);
    return FRP_OK;
}

}  // namespace frp
