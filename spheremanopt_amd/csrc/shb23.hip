// SHB23 (1-D Chebyshev Swift-Hohenberg) — placeholder until the kernels land.
#include "smo_common.hpp"
namespace smo {
Context* make_shb23(const smo_config&) { set_error("SHB23 device path not built yet"); return nullptr; }
}  // namespace smo
