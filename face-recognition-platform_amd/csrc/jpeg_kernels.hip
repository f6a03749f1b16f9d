// Device half of the JPEG ingest: quantised DCT coefficients (decoded from the bit stream on the host,
// jpeg_host.cpp) -> dequantisation -> inverse DCT -> chroma upsampling -> YCbCr to BGR, written straight into the staging
// frame buffer the detector's stem reads ([B,H,W,3] u8 BGR).
//
// Replaces the pixel half of the PIL decode behind `face_recognition.load_image_file` (backend/app/services/face_service.py:139,
// backend/app/routes/face.py:177-185,216,404,976).  PIL decodes with libjpeg(-turbo) at its defaults, so the three steps
// follow those algorithms as published, in integer arithmetic, and reproduce them BIT FOR BIT (tests: device vs PIL on
// committed and generated stills, max difference 0):
//   * inverse DCT: the "slow" integer algorithm (Loeffler-Ligtenberg-Moschytz, 13-bit constants, columns then rows, the
//     column pass keeping 2 extra fraction bits), + 128, clamp;
//   * chroma upsampling: the "fancy" triangle filters - 4:2:0: 3/4 nearer + 1/4 farther row, then the same along the row
//     with rounding terms 8 / 7; 4:2:2: along the row with rounding terms 1 / 2; edge samples replicate, and only the REAL
//     extent of the chroma planes (ceil(W / 2) x ceil(H / 2)) takes part, not their padding up to whole MCUs;
//   * colour: R = Y + 1.402 Cr', G = Y - 0.34414 Cb' - 0.71414 Cr', B = Y + 1.772 Cb' in 16-bit fixed point with the
//     rounding of the table form (the G term sums both products before the shift).
// Both kernels are HBM-bound and small next to the pipeline (2 B read per coefficient, 1 B written per sample; 1.5 B read and
// 3 B written per pixel): they run on the handle's COPY stream, under the previous batch's kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "frp_internal.h"

namespace frp {

namespace {

#define JC_BITS 13
#define JP1_BITS 2
#define JF_0_298631336 2446
#define JF_0_390180644 3196
#define JF_0_541196100 4433
#define JF_0_765366865 6270
#define JF_0_899976223 7373
#define JF_1_175875602 9633
#define JF_1_501321110 12299
#define JF_1_847759065 15137
#define JF_1_961570560 16069
#define JF_2_053119869 16819
#define JF_2_562915447 20995
#define JF_3_072711026 25172

// one LL&M pass on 8 values; outputs descaled by SHIFT (arithmetic shift, round half up)
template <int SHIFT>
__device__ __forceinline__ void idct8(const int (&in)[8], int (&out)[8]) {
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * JF_0_541196100;
    const int tmp2 = z1 + z3 * (-JF_1_847759065);
    const int tmp3 = z1 + z2 * JF_0_765366865;
    const int tmp0 = (in[0] + in[4]) << JC_BITS;
    const int tmp1 = (in[0] - in[4]) << JC_BITS;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int t0 = in[7], t1 = in[5], t2 = in[3], t3 = in[1];
    z1 = t0 + t3;
    z2 = t1 + t2;
    z3 = t0 + t2;
    int z4 = t1 + t3;
    const int z5 = (z3 + z4) * JF_1_175875602;
    t0 *= JF_0_298631336;
    t1 *= JF_2_053119869;
    t2 *= JF_3_072711026;
    t3 *= JF_1_501321110;
    z1 *= -JF_0_899976223;
    z2 *= -JF_2_562915447;
    z3 = z3 * (-JF_1_961570560) + z5;
    z4 = z4 * (-JF_0_390180644) + z5;
    t0 += z1 + z3;
    t1 += z2 + z4;
    t2 += z2 + z3;
    t3 += z1 + z4;
    constexpr int R = 1 << (SHIFT - 1);
    out[0] = (tmp10 + t3 + R) >> SHIFT;
    out[7] = (tmp10 - t3 + R) >> SHIFT;
    out[1] = (tmp11 + t2 + R) >> SHIFT;
    out[6] = (tmp11 - t2 + R) >> SHIFT;
    out[2] = (tmp12 + t1 + R) >> SHIFT;
    out[5] = (tmp12 - t1 + R) >> SHIFT;
    out[3] = (tmp13 + t0 + R) >> SHIFT;
    out[4] = (tmp13 - t0 + R) >> SHIFT;
}

// 32 blocks per 256-thread workgroup, 8 threads per block: row load + dequantise -> LDS, column pass, row pass -> 8 samples
__global__ __launch_bounds__(256) void jpeg_idct_kernel(JpegParams p) {
    __shared__ int ws[32][8][9];               // [block][row][column], padded
    const int t = threadIdx.x, bl = t >> 3, k = t & 7;
    const long blk = (long)blockIdx.x * 32 + bl;
    const bool live = blk < (long)p.B * p.blocks_per_image;
    int c = 0, by = 0, bx = 0, b = 0;
    if (live) {
        b = (int)(blk / p.blocks_per_image);
        int r = (int)(blk - (long)b * p.blocks_per_image);
        while (c + 1 < p.components && r >= p.bx[c] * p.by[c]) { r -= p.bx[c] * p.by[c]; ++c; }
        by = r / p.bx[c];
        bx = r - by * p.bx[c];
        const int16_t* src = p.coef + blk * 64 + k * 8;
        const uint16_t* q = p.qtab + ((long)b * 3 + c) * 64 + k * 8;
        const uint4 raw = *reinterpret_cast<const uint4*>(src);
        const uint4 qq = *reinterpret_cast<const uint4*>(q);
        const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w}, qw[4] = {qq.x, qq.y, qq.z, qq.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ws[bl][k][2 * i] = (int)(int16_t)(rw[i] & 0xffffu) * (int)(qw[i] & 0xffffu);
            ws[bl][k][2 * i + 1] = (int)(int16_t)(rw[i] >> 16) * (int)(qw[i] >> 16);
        }
    }
    __syncthreads();
    int v[8], o[8];
    if (live) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = ws[bl][r][k];                 // column k
        idct8<JC_BITS - JP1_BITS>(v, o);
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[bl][r][k] = o[r];
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ws[bl][k][i];                 // row k
        idct8<JC_BITS + JP1_BITS + 3>(v, o);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int a = o[i] + 128, d = o[i + 4] + 128;
            a = a < 0 ? 0 : (a > 255 ? 255 : a);
            d = d < 0 ? 0 : (d > 255 ? 255 : d);
            lo |= (unsigned)a << (8 * i);
            hi |= (unsigned)d << (8 * i);
        }
        uint8_t* plane = p.planes + p.plane_off[c] + (long)b * p.plane_img;
        *reinterpret_cast<uint2*>(plane + (long)(by * 8 + k) * (p.bx[c] * 8) + bx * 8) = make_uint2(lo, hi);
    }
}

__device__ __forceinline__ int chroma_h2v2(const uint8_t* pl, int pitch, int cw, int ch, int x, int y) {
    const int cy = y >> 1, cx = x >> 1;
    const int fy = (y & 1) ? (cy + 1 < ch ? cy + 1 : cy) : (cy > 0 ? cy - 1 : cy);
    const int s = 3 * pl[(long)cy * pitch + cx] + pl[(long)fy * pitch + cx];
    if (x & 1) {
        if (cx == cw - 1) return (4 * s + 7) >> 4;
        const int n = 3 * pl[(long)cy * pitch + cx + 1] + pl[(long)fy * pitch + cx + 1];
        return (3 * s + n + 7) >> 4;
    }
    if (cx == 0) return (4 * s + 8) >> 4;
    const int l = 3 * pl[(long)cy * pitch + cx - 1] + pl[(long)fy * pitch + cx - 1];
    return (3 * s + l + 8) >> 4;
}

__device__ __forceinline__ int chroma_h2v1(const uint8_t* pl, int pitch, int cw, int x, int y) {
    const int cx = x >> 1;
    const int v = pl[(long)y * pitch + cx];
    if (x & 1) return cx == cw - 1 ? v : (3 * v + pl[(long)y * pitch + cx + 1] + 2) >> 2;
    return cx == 0 ? v : (3 * v + pl[(long)y * pitch + cx - 1] + 1) >> 2;
}

// one thread per output pixel: upsampled chroma + colour conversion -> 3 bytes of the BGR frame
__global__ __launch_bounds__(256) void jpeg_color_kernel(JpegParams p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)p.W * p.H;
    if (i >= per * p.B) return;
    const int b = (int)(i / per);
    const int rem = (int)(i - (long)b * per);
    const int y = rem / p.W, x = rem - y * p.W;
    const uint8_t* base = p.planes + (long)b * p.plane_img;
    const int yv = base[p.plane_off[0] + (long)y * (p.bx[0] * 8) + x];
    uint8_t* dst = p.frames + i * 3;
    if (p.components == 1) { dst[0] = dst[1] = dst[2] = (uint8_t)yv; return; }
    const int pitch = p.bx[1] * 8;
    const uint8_t* pcb = base + p.plane_off[1];
    const uint8_t* pcr = base + p.plane_off[2];
    int cb, cr;
    if (p.hs == 2 && p.vs == 2) {
        cb = chroma_h2v2(pcb, pitch, p.cw, p.ch, x, y);
        cr = chroma_h2v2(pcr, pitch, p.cw, p.ch, x, y);
    } else if (p.hs == 2) {
        cb = chroma_h2v1(pcb, pitch, p.cw, x, y);
        cr = chroma_h2v1(pcr, pitch, p.cw, x, y);
    } else {
        cb = pcb[(long)y * pitch + x];
        cr = pcr[(long)y * pitch + x];
    }
    cb -= 128;
    cr -= 128;
    int r = yv + ((91881 * cr + 32768) >> 16);
    int g = yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    int bl = yv + ((116130 * cb + 32768) >> 16);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    bl = bl < 0 ? 0 : (bl > 255 ? 255 : bl);
    dst[0] = (uint8_t)bl;
    dst[1] = (uint8_t)g;
    dst[2] = (uint8_t)r;
}


// -------------------------------------------------------------------------------------------------------------------------
// Entropy decoding on the device (round 5): one THREAD per restart interval.  The DC predictors and the byte alignment reset at
// every RSTn marker (T.81 F.1.1.5), so a scan with restart intervals is a set of independent bit streams: a 1080p 4:2:0 frame
// with one interval per MCU row is 68 of them, a batch of 32 frames 2,176 threads.  Each walks its bytes (un-stuffing 0xFF00),
// decodes with the image's own tables (copied to LDS: one workgroup = 64 intervals of ONE image) and scatters the non-zero
// coefficients into the buffer the inverse-DCT kernel reads (zeroed before the launch).  The same canonical decoding as
// jpeg_host.cpp - bit-identical coefficients (tests/test_gpu_pipeline.py: both paths against PIL).  What moves over PCIe is the
// compressed stream (~0.5 MB per 1080p frame) instead of 6.3 MB of coefficients, and the host keeps its threads.
struct DevBits {
    const unsigned* wp;           // next aligned dword of the interval's bytes
    unsigned long long raw;       // bytes fetched but not yet fed (next byte = bits 0..7)
    int rawn;                     // ... how many
    int left;                     // bytes of the interval not yet fed into the window (incl. those in `raw`)
    unsigned long long acc;       // next bit of the stream = bit 63
    int nbits;
    int pad;                      // zero bits fed behind the end of the interval
    // (bytes come in 8 at a time - two aligned dword loads -: fetched one by one, every lane of a wave waited a memory round trip
    // per byte of its own stream and a batch of 32 x 1080p took 18 ms)
    __device__ __forceinline__ void init(const uint8_t* p, const uint8_t* end) {
        const unsigned long a = (unsigned long)p;
        wp = reinterpret_cast<const unsigned*>(a & ~3ul);
        const int skip_ = (int)(a & 3ul);
        raw = (unsigned long long)(*wp++) >> (8 * skip_);
        rawn = 4 - skip_;
        left = (int)(end - p);
        acc = 0;
        nbits = 0;
        pad = 0;
    }
    __device__ __forceinline__ unsigned next_byte() {          // only while left > 0
        if (rawn == 0) {
            const unsigned lo = wp[0], hi = wp[1];             // (reads up to 7 bytes past the interval: inside the staging buffer's padding)
            wp += 2;
            raw = (unsigned long long)lo | ((unsigned long long)hi << 32);
            rawn = 8;
        }
        const unsigned b = (unsigned)(raw & 0xffu);
        raw >>= 8;
        --rawn;
        --left;
        return b;
    }
    __device__ __forceinline__ void fill() {
        while (nbits <= 56) {
            unsigned b = 0;
            if (left > 0) {
                b = next_byte();
                if (b == 0xFF) {                        // a stuffed zero follows; anything else is a marker INSIDE the interval: end of data
                    unsigned n = 1;
                    if (left > 0) n = next_byte();
                    if (n != 0x00) { left = 0; b = 0; pad += 8; }
                }
            } else {
                pad += 8;
            }
            acc |= (unsigned long long)b << (56 - nbits);
            nbits += 8;
        }
    }
    __device__ __forceinline__ unsigned peek(int n) const { return (unsigned)(acc >> (64 - n)); }
    __device__ __forceinline__ void skip(int n) { acc <<= n; nbits -= n; }
    __device__ __forceinline__ int extend(int s) {
        if (s == 0) return 0;
        const int v = (int)peek(s);
        skip(s);
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    __device__ __forceinline__ int decode(const JpegHuffTableDev& t) {
        const unsigned f = t.fast[peek(9)];
        if (f) { skip((int)(f >> 8)); return (int)(f & 0xff); }
        for (int len = 10; len <= 16; ++len) {
            const int code = (int)peek(len);
            if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len]) {
                skip(len);
                return t.vals[t.valptr[len] + code - t.mincode[len]];
            }
        }
        return -1;
    }
};

__constant__ uint8_t kZigZagDev[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                       41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                       30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__global__ __launch_bounds__(64) void jpeg_huffman_kernel(JpegHuffParams p) {
    __shared__ JpegHuffTableDev tab[6];

    __shared__ uint8_t zz[64];                 // (in LDS: indexed per lane - from constant memory every symbol waited a vector-memory round trip)
    zz[threadIdx.x] = kZigZagDev[threadIdx.x];
    const int b = blockIdx.y;
    {   // this image's tables -> LDS (dwords: the struct is a multiple of 4 bytes)
        const unsigned* src = reinterpret_cast<const unsigned*>(p.tables + (long)b * 6);
        unsigned* dst = reinterpret_cast<unsigned*>(tab);
        for (int i = threadIdx.x; i < (int)(6 * sizeof(JpegHuffTableDev) / 4); i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const int it = blockIdx.x * 64 + threadIdx.x;
    if (it >= p.n_int) return;
    const uint32_t* io = p.int_off + (long)b * (p.n_int + 1);
    DevBits br;
    // an interval ends where the next one's marker starts (2 bytes in front of the next interval's first byte); the last one at the end of the data
    br.init(p.scan + io[it], p.scan + (it + 1 < p.n_int ? io[it + 1] - 2 : io[p.n_int]));
    int16_t* coef = p.coef + (long)b * p.coef_per_image;
    const long total = (long)p.mcus_x * p.mcus_y;
    long m0 = (long)it * p.ri, m1 = m0 + p.ri;
    m1 = m1 < total ? m1 : total;
    int pred0 = 0, pred1 = 0, pred2 = 0;
    bool bad = false;
    for (long m = m0; m < m1 && !bad; ++m) {
        const int my = (int)(m / p.mcus_x), mx = (int)(m - (long)my * p.mcus_x);
        for (int c = 0; c < p.components && !bad; ++c) {
            const JpegHuffTableDev& dct = tab[2 * c];
            const JpegHuffTableDev& act = tab[2 * c + 1];
            for (int v = 0; v < p.vs[c] && !bad; ++v)
                for (int hh = 0; hh < p.hs[c] && !bad; ++hh) {
                    int16_t* blk = coef + p.comp_off[c] + ((long)(my * p.vs[c] + v) * p.bx[c] + (mx * p.hs[c] + hh)) * 64;
                    br.fill();
                    const int s = br.decode(dct);
                    if (s < 0 || s > 11) { bad = true; break; }
                    const int diff = br.extend(s);
                    int pr = c == 0 ? pred0 : (c == 1 ? pred1 : pred2);
                    pr += diff;
                    if (c == 0) pred0 = pr; else if (c == 1) pred1 = pr; else pred2 = pr;
                    if (pr) blk[0] = (int16_t)pr;
                    for (int k = 1; k < 64;) {
                        br.fill();
                        // (a combined code + magnitude look-up as in the host decoder, and refilling only below 32 bits, made this loop
                        // SLOWER - 32 ms against 18 for 32 x 1080p: the lanes of a wave walk 64 different streams and every extra branch is
                        // executed by all of them)
                        const int rs = br.decode(act);
                        if (rs < 0) { bad = true; break; }
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r == 15) { k += 16; continue; }
                            break;                                   // end of block
                        }
                        k += r;
                        if (k > 63) { bad = true; break; }
                        blk[zz[k]] = (int16_t)br.extend(sz);
                        ++k;
                    }
                }
        }
    }
    // bits that were not in the interval were consumed: it ends before its last MCU (jpeg_host.cpp: ran_dry)
    if (bad || br.nbits < br.pad) atomicOr(p.err + b, 1);
}

}  // namespace

hipError_t launch_jpeg_decode(const JpegParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.W <= 0 || p.H <= 0 || p.blocks_per_image <= 0 || !p.coef || !p.qtab || !p.planes || !p.frames) return hipErrorInvalidValue;
    const long blocks = (long)p.B * p.blocks_per_image;
    const long pixels = (long)p.B * p.W * p.H;
    if ((blocks + 31) / 32 > 0x7fffffffL || (pixels + 255) / 256 > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((blocks + 31) / 32)), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_jpeg_huffman(const JpegHuffParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.n_int <= 0 || p.ri <= 0 || p.components <= 0 || p.components > 3 || !p.scan || !p.int_off || !p.tables || !p.coef || !p.err)
        return hipErrorInvalidValue;
    if (p.B > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(jpeg_huffman_kernel, dim3((unsigned)((p.n_int + 63) / 64), (unsigned)p.B), dim3(64), 0, stream, p);
    return hipGetLastError();
}

}  // namespace frp
