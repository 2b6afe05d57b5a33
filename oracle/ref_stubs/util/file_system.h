// Stand-in for MVE's util/file_system.h (absent).  histogram.cpp / sparse_table.h include it without using it; generate_texture_views.cpp
// uses a directory listing (name, is_dir, absolute name; sortable by name), path joins and file tests: restated here on POSIX calls.
// Test infrastructure only (oracle/_ref).
#ifndef MVS_REF_STUB_UTIL_FILE_SYSTEM_H
#define MVS_REF_STUB_UTIL_FILE_SYSTEM_H
#include <dirent.h>
#include <sys/stat.h>
#include <climits>
#include <cstdlib>
#include <string>
#include <vector>
#include "util/exception.h"
namespace util { namespace fs {
inline std::string join_path(std::string const& a, std::string const& b) { if (a.empty()) return b; return (a[a.size() - 1] == '/') ? a + b : a + "/" + b; }
inline std::string basename(std::string const& p) { const std::size_t k = p.find_last_of('/'); return k == std::string::npos ? p : p.substr(k + 1); }
inline std::string abspath(std::string const& p) { char buf[PATH_MAX]; return realpath(p.c_str(), buf) ? std::string(buf) : p; }
inline std::string replace_extension(std::string const& fn, std::string const& ext) {
    const std::size_t slash = fn.find_last_of('/'), dot = fn.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return fn + "." + ext;
    return fn.substr(0, dot) + "." + ext;
}
inline bool file_exists(char const* p) { struct stat st; return ::stat(p, &st) == 0 && S_ISREG(st.st_mode); }
inline bool dir_exists(char const* p) { struct stat st; return ::stat(p, &st) == 0 && S_ISDIR(st.st_mode); }
struct File {
    std::string path, name; bool is_dir;
    File() : is_dir(false) {}
    File(std::string const& p, std::string const& n, bool d) : path(p), name(n), is_dir(d) {}
    std::string get_absolute_name() const { return join_path(path, name); }
    bool operator<(File const& rhs) const { return name < rhs.name; }     // (files of ONE directory: ordered by name)
};
class Directory : public std::vector<File> {
public:
    Directory() {}
    explicit Directory(std::string const& path) { scan(path); }
    void scan(std::string const& path) {
        this->clear();
        DIR* d = ::opendir(path.c_str());
        if (!d) throw util::Exception("cannot open directory " + path);
        while (struct dirent* e = ::readdir(d)) {
            const std::string n(e->d_name);
            if (n == "." || n == "..") continue;
            this->push_back(File(path, n, dir_exists(join_path(path, n).c_str())));
        }
        ::closedir(d);
    }
};
} }
#endif
