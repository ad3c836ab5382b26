// Thread-local error string behind crt_last_error() (include/crt.h).
#pragma once
#include <string>
namespace crt {
extern thread_local std::string g_last_error;
int fail(int code, const std::string& msg);
}  // namespace crt
