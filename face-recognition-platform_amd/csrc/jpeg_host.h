// Host half of the JPEG ingest (jpeg_host.cpp): header parsing + Huffman decoding of baseline JPEG stills.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "frp.h"

namespace frp {

size_t jpeg_coef_elems(const frp_jpeg_info& info);
int jpeg_info(const uint8_t* data, size_t size, frp_jpeg_info* out, std::string* err);
// coef: per component [blocks_y][blocks_x][64] int16 (natural order, quantised), components back to back; qtab: [3][64] uint16
int jpeg_decode_coefficients(const uint8_t* data, size_t size, int16_t* coef, size_t coef_elems, uint16_t* qtab, frp_jpeg_info* info,
                             std::string* err);

// What the device entropy decoder needs of one image (frp_upload_jpeg_async's device path): headers parsed as above, the scan
// located, the restart markers found and checked (count and RST0..7 sequence).  Fails (FRP_ERR_INVALID) for files without
// restart intervals and for anything the host decoder refuses.
// Canonical Huffman table in the form both decoders walk (T.81 F.2.2.3) + a 9-bit look-ahead: (code length << 8) | value, 0 = a
// longer code.  (Plain data: shared with the device code through frp_internal.h.)
struct JpegHuffTableDev {
    uint16_t fast[512];
    int32_t mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
};
struct JpegDevicePlan {
    frp_jpeg_info info;
    uint16_t qtab[192];                 // [3][64] natural order
    const uint8_t* scan = nullptr;      // first entropy-coded byte
    size_t scan_bytes = 0;              // up to the end of the last interval (the marker behind it, or the end of the file)
    std::vector<uint32_t> int_off;      // [n_int + 1] offsets from `scan`
};
int jpeg_plan_device_decode(const uint8_t* data, size_t size, JpegDevicePlan& plan, JpegHuffTableDev* tables6, std::string* err);

}  // namespace frp
