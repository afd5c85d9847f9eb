// Process-wide device context and error translation for the host mirror.
#ifndef KRYLOV_HOST_DEVICE_HPP_
#define KRYLOV_HOST_DEVICE_HPP_

#include <cstdlib>
#include <stdexcept>
#include <string>

#include "krylov_amd.h"

namespace krylov {

inline void check(int status, const char *what) {
  if (status != KRY_OK)
    throw std::runtime_error(std::string(what) + ": " + kry_last_error() +
                             " (krylov_amd status " + std::to_string(status) + ")");
}

// One context per process, on device $KRYLOV_DEVICE (default 0).  There is no
// CPU fallback: without a HIP device this throws.
inline kry_ctx *default_context() {
  static kry_ctx *ctx = [] {
    const char *env = std::getenv("KRYLOV_DEVICE");
    kry_ctx *c = nullptr;
    check(kry_ctx_create(env ? std::atoi(env) : 0, nullptr, &c), "kry_ctx_create");
    return c;
  }();
  return ctx;
}

}  // namespace krylov

#endif  // KRYLOV_HOST_DEVICE_HPP_
