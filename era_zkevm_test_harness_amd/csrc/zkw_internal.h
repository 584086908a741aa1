// zkw_internal.h — the little that the library's translation units share besides include/zkw.h.
#pragma once
// sets the calling thread's zkw_last_error() text and returns `code` (defined in zkw_api.hip)
int zkw_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
