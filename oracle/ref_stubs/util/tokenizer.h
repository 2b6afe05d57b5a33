// Stand-in for MVE's util/tokenizer.h (absent).  DEFINED HERE from recollection: a vector of strings; split(str, delim = ' ',
// keep_empty = false) cuts at the delimiter character and drops empty tokens; concat(pos, num) joins tokens [pos, pos + num) -- num = 0:
// to the end -- with single blanks.  Test infrastructure only (oracle/_ref: the reference's .cam parsing in generate_texture_views.cpp).
#ifndef MVS_REF_STUB_UTIL_TOKENIZER_H
#define MVS_REF_STUB_UTIL_TOKENIZER_H
#include <string>
#include <vector>
#include "util/string.h"
namespace util {
class Tokenizer : public std::vector<std::string> {
public:
    void split(std::string const& str, char delim = ' ', bool keep_empty = false) {
        this->clear();
        std::size_t last = 0;
        for (std::size_t i = 0; i <= str.size(); ++i) {
            if (i == str.size() || str[i] == delim) {
                if (keep_empty || i > last) this->push_back(str.substr(last, i - last));
                last = i + 1;
            }
        }
    }
    std::string concat(std::size_t pos, std::size_t num = 0) const {
        std::string out;
        const std::size_t end = (num == 0) ? this->size() : std::min(this->size(), pos + num);
        for (std::size_t i = pos; i < end; ++i) { if (i > pos) out += ' '; out += (*this)[i]; }
        return out;
    }
};
}  // namespace util
#endif
