// Stand-in for MVE's mve/image_io.h.  No files are involved in oracle/_ref: the "file name" a TextureView is constructed
// with is "<width>x<height>", images are attached with TextureView::bind_image.
#ifndef MVS_REF_STUB_MVE_IMAGE_IO_H
#define MVS_REF_STUB_MVE_IMAGE_IO_H
#include <cstdio>
#include <cstdlib>
#include <string>
#include "mve/image.h"
#include "util/exception.h"
namespace mve { namespace image {
struct ImageHeaders { int width, height, channels; };
inline ImageHeaders load_file_headers(std::string const& name) {
    ImageHeaders h; h.channels = 3;
    if (std::sscanf(name.c_str(), "%dx%d", &h.width, &h.height) != 2) throw util::Exception("stand-in image name must be <w>x<h>");
    return h;
}
inline ByteImage::Ptr load_file(std::string const&) { throw util::Exception("oracle/_ref never loads image files"); }
inline void save_png_file(ByteImage::Ptr, std::string const&) {}
} }  // namespace mve::image
#endif
