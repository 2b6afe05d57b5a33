// Stand-in for MVE's util/string.h: the three helpers the reference's generate_texture_views.cpp uses (MVE is an un-vendored download;
// their semantics -- last / first n characters, ASCII upper case -- are restated from their names and uses; test infrastructure only).
#ifndef MVS_REF_STUB_UTIL_STRING_H
#define MVS_REF_STUB_UTIL_STRING_H
#include <cctype>
#include <string>
namespace util { namespace string {
inline std::string left(std::string const& s, std::size_t n) { return s.substr(0, n < s.size() ? n : s.size()); }
inline std::string right(std::string const& s, std::size_t n) { return n >= s.size() ? s : s.substr(s.size() - n); }
inline std::string uppercase(std::string const& s) { std::string r(s); for (std::size_t i = 0; i < r.size(); ++i) r[i] = (char)std::toupper((unsigned char)r[i]); return r; }
} }
#endif
