// Host-side helpers shared by every translation unit of libsfcvit_hip.so.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/sfcvit.h"

namespace sfcvit {

// Records the message for sfcvit_last_error() and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();
// hipGetLastError() -> SFCVIT_ELAUNCH with the HIP message, or SFCVIT_OK.
int check_launch(const char *what);

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace sfcvit
