// polympc_amd — kernels and host entry points of the built-in OCP KiteStandInOCP (one translation unit per model: see pmpc_builtin.hpp)
#define PMPC_BUILTIN_DEFINITIONS
#include "pmpc_builtin.hpp"
namespace pmpc { template <> struct LDS_PATH_PROFILED<KiteStandInOCP> { static constexpr bool value = true; }; }   // large-instance kernel also built with phase timers (PMPC_PHASE_PROFILE=1)
PMPC_INSTANTIATE_BUILTIN(pmpc::KiteStandInOCP)
