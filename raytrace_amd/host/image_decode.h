// image_decode.h - file decoders behind rt::ImageTexture (raytrace.cpp).
#pragma once
#include <string>
#include <vector>

namespace rt {
namespace detail {

// Sequential Huffman JPEG (jpeg_baseline.cpp).  On success w x h and w * h * 3 bytes of RGB,
// row 0 = top - the texels the reference's loader produces for the same file; on failure
// `why` says what is wrong with the file or which JPEG feature is not supported.
bool load_jpeg(const std::vector<unsigned char>& data, int& w, int& h, std::vector<unsigned char>& rgb, std::string& why);

}  // namespace detail
}  // namespace rt
