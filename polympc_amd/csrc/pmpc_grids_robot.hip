// polympc_amd — register-resident SQP kernels of RobotOCP for the 3-, 4-, 6-, 8-, 9-, 10-, 12- and 13-node grids (see pmpc_grids.hpp)
#include "pmpc_grids.hpp"
PMPC_INSTANTIATE_GRIDS(pmpc::RobotOCP)
