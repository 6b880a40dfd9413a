// sks_err.h -- error reporting shared by the translation units outside sks_raster.hip: the text goes into
// sks_raster.hip's thread-local buffer, which sks_last_error() returns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

extern "C" void sks_set_error_(const char* msg);

namespace {

thread_local char g_err2[512] = "";

[[maybe_unused]] int fail2(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err2, sizeof(g_err2), fmt, ap);
    va_end(ap);
    sks_set_error_(g_err2);
    return code;
}

}  // namespace

#define HIP_TRY2(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail2((int)e_, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)
