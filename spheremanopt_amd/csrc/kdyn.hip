// KDYN (3-D kinematic dynamo) — placeholder until the kernels land.
#include "smo_common.hpp"
namespace smo {
Context* make_kdyn(const smo_config&) { set_error("KDYN device path not built yet"); return nullptr; }
}  // namespace smo
