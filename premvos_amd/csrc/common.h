// Shared helpers for libpremvos_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/premvos_hip.h"

namespace premvos {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PREMVOS_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return PREMVOS_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace premvos

#define PV_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) return premvos::fail(PREMVOS_EINVAL, __VA_ARGS__); \
  } while (0)
