// Host half of the JPEG ingest (jpeg_host.cpp): header parsing + Huffman decoding of baseline JPEG stills.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

#include "frp.h"

namespace frp {

size_t jpeg_coef_elems(const frp_jpeg_info& info);
int jpeg_info(const uint8_t* data, size_t size, frp_jpeg_info* out, std::string* err);
// coef: per component [blocks_y][blocks_x][64] int16 (natural order, quantised), components back to back; qtab: [3][64] uint16
int jpeg_decode_coefficients(const uint8_t* data, size_t size, int16_t* coef, size_t coef_elems, uint16_t* qtab, frp_jpeg_info* info,
                             std::string* err);

}  // namespace frp
