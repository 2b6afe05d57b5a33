// Stand-in for MVE's util/exception.h (MVE is an un-vendored download): the only thing the reference's
// self-contained sources (histogram.cpp, sparse_table.h) use from it are the two exception types below.  This file lets
// those sources compile UNCHANGED from /root/reference for oracle/_ref (test infrastructure; see oracle/Makefile `ref`).
#ifndef MVS_REF_STUB_UTIL_EXCEPTION_H
#define MVS_REF_STUB_UTIL_EXCEPTION_H
#include <exception>
#include <string>
namespace util {
class Exception : public std::exception {
public:
    explicit Exception(std::string const& msg = std::string()) : text(msg) {}
    virtual ~Exception() throw() {}
    virtual const char* what() const throw() { return text.c_str(); }
protected:
    std::string text;
};
class FileException : public Exception {
public:
    FileException(std::string const& filename, std::string const& msg) : Exception(filename + ": " + msg), filename(filename) {}
    virtual ~FileException() throw() {}
    std::string filename;
};
}  // namespace util
#endif
