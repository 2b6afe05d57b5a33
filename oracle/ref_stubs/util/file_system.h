// Stand-in for MVE's util/file_system.h: included by the reference's histogram.cpp / sparse_table.h, none of its
// functions are used by them.  See util/exception.h in this directory.
#ifndef MVS_REF_STUB_UTIL_FILE_SYSTEM_H
#define MVS_REF_STUB_UTIL_FILE_SYSTEM_H
#include <string>
namespace util { namespace fs {} }
#endif
