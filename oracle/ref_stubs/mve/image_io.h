// Stand-in for MVE's mve/image_io.h.  No files are involved in oracle/_ref: the "file name" a TextureView is constructed
// with starts with "<width>x<height>"; images are attached with TextureView::bind_image, or -- for the reference's own
// load_image() call in calculate_data_costs.cpp:157 -- looked up by that name in a registry the test fills.
#ifndef MVS_REF_STUB_MVE_IMAGE_IO_H
#define MVS_REF_STUB_MVE_IMAGE_IO_H
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "mve/image.h"
#include "util/exception.h"
namespace mve { namespace image {
struct ImageHeaders { int width, height, channels; };
inline std::vector<std::string>& header_requests() { static std::vector<std::string> l; return l; }   // every name load_file_headers was asked for (a TextureView's constructor asks once)
inline ImageHeaders load_file_headers(std::string const& name) {
    header_requests().push_back(name);
    ImageHeaders h; h.channels = 3;
    const std::size_t slash = name.find_last_of('/');                      // (a path: the file's own name carries the size)
    if (std::sscanf(name.c_str() + (slash == std::string::npos ? 0 : slash + 1), "%dx%d", &h.width, &h.height) != 2) throw util::Exception("stand-in image name must start with <w>x<h>");
    return h;
}
inline std::map<std::string, ByteImage::Ptr>& file_registry() { static std::map<std::string, ByteImage::Ptr> r; return r; }
inline bool& load_copies() { static bool b = false; return b; }                        // true: load_file returns a fresh COPY of the registered image (as decoding a file does)
inline std::set<std::string>& unreadable_files() { static std::set<std::string> s; return s; }   // names load_file refuses (a file that went missing)
inline ByteImage::Ptr load_file(std::string const& name) {
    if (unreadable_files().count(name)) throw util::Exception("Cannot open file: " + name);
    std::map<std::string, ByteImage::Ptr>::const_iterator it = file_registry().find(name);
    if (it != file_registry().end()) return load_copies() ? ByteImage::Ptr(new ByteImage(*it->second)) : it->second;
    ImageHeaders h;
    try { h = load_file_headers(name); header_requests().pop_back(); } catch (util::Exception&) { header_requests().pop_back(); throw util::Exception("oracle/_ref: no image registered as " + name); }
    return ByteImage::create(h.width, h.height, 3);                        // an unregistered file whose name states its size: blank pixels
}
inline std::vector<std::string>& saved_files() { static std::vector<std::string> l; return l; }   // names handed to save_png_file, in call order
inline void save_png_file(ByteImage::Ptr, std::string const& name) { saved_files().push_back(name); }
} }  // namespace mve::image
#endif
