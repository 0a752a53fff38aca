// fd_host.h -- shared by the host translation units of libfastdiff_hip.so (fd_api.cpp, fd_api_ext.cpp, fd_api_train.cpp): the error
// macros and the few helpers of the core that the other two call.
#pragma once
#include <stdio.h>

#include <string>

#include "fd_internal.h"

extern std::string g_create_error;      // text of a failed fd_create (no handle to keep it)

#define FD_FAIL(h, code, ...)                                   \
    do {                                                        \
        char buf__[512];                                        \
        snprintf(buf__, sizeof(buf__), __VA_ARGS__);            \
        if (h) (h)->err = buf__; else g_create_error = buf__;   \
        return (code);                                          \
    } while (0)

#define FD_HIP(h, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) FD_FAIL(h, FD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__));    \
    } while (0)

extern "C" {      // (defined inside fd_api.cpp's extern "C" block; hidden visibility: not part of the ABI)
// fallback = host: look at the flags of a pending fd_sample before touching device state (no-op when nothing is pending)
int fd_settle(fd_handle h);
// Pinned staging ring (fd_context::stage): the next slot with room for `bytes`, free to be written by the host; ... and the mark behind
// the copies that read it
int fd_stage_acquire(fd_handle h, size_t bytes, fd_context::StageSlot **out);
int fd_stage_commit(fd_handle h, fd_context::StageSlot *sl, hipStream_t stream);
}
void fd_prof_drain(fd_context *c);
