#!/bin/bash
# round 6 (late): the problems / policy set of the reference's control tests on the condensed register kernel (default) against the full two-rows-per-lane inverse they ran on before
cd $GRAFT_REPO_ROOT
echo "== minimal_time_test.cpp (NP = 1), 4096 perturbed instances and a lone instance: default, then PMPC_NO_CONDREG=1"
python tests/experiments/parking_np1_bench.py; PMPC_NO_CONDREG=1 python tests/experiments/parking_np1_bench.py
B=1 python tests/experiments/parking_np1_bench.py; B=1 PMPC_NO_CONDREG=1 python tests/experiments/parking_np1_bench.py
echo "== nonlinear_constraints_test.cpp (NP = 1, NG = 1): bound 1.2 (binding) and 10 (the reference's, inactive)"
for u in 1.2 10.0; do UBG=$u python tests/experiments/parking_ng_bench.py; UBG=$u PMPC_NO_CONDREG=1 python tests/experiments/parking_ng_bench.py; done
B=1 python tests/experiments/parking_ng_bench.py; B=1 PMPC_NO_CONDREG=1 python tests/experiments/parking_ng_bench.py
echo "== policy hooks on robot grids (valet_parking_mpc_test.cpp's set; Ruiz alone; filter alone; mirroring): default, then PMPC_NO_CONDREG_RUIZ=1"
python tests/experiments/ruiz_condreg_bench.py; PMPC_NO_CONDREG_RUIZ=1 python tests/experiments/ruiz_condreg_bench.py 2>&1 | grep -v "regularisation\|{'line_search': 1}"
echo "== eigenvalue mirroring, cost per SQP iteration"
python tests/experiments/mirroring_cost_probe.py
