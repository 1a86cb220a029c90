// polympc_amd — kernels and host entry points of the built-in OCP KiteStandInOCP (one translation unit per model: see pmpc_builtin.hpp)
#define PMPC_BUILTIN_DEFINITIONS
#include "pmpc_builtin.hpp"
PMPC_INSTANTIATE_BUILTIN(pmpc::KiteStandInOCP)
