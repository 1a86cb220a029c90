import sys; sys.path.insert(0,'/root/repo')
import numpy as np, polympc_amd as pa
from polympc_amd import workloads
from oracle import binding as ob
B=8; wl=workloads.robot_batch(B); ctx=pa.Context(0)
import os
ss=pa.sqp_settings_default(); ss.max_iter=10; ss.line_search_max_iter=10
if os.environ.get('SERIAL_LS'): ss.rho=-1.0
x,lam,info=ctx.sqp_solve_batch(0,6,1,0.,2.,B,wl['d'],wl['lbx'],wl['ubx'],sqp_settings=ss)
oss=ob.sqp_default_settings(); oss.max_iter=10; oss.line_search_max_iter=10
xo,lo,io=ob.sqp_solve_batch(0,6,1,0.,2.,B,wl['d'],wl['lbx'],wl['ubx'],sqp_settings=oss,pivot=1)
print('gpu iter',info['iter'],'qpit',info['qp_solver_iter'],'st',info['status'])
print('orc iter',[i.iter for i in io],'qpit',[i.qp_solver_iter for i in io])
print('dx',np.abs(x-xo).max(axis=1)); print('cost',info['cost'],[i.cost for i in io]); print('viol',info['max_violation'],[i.max_violation for i in io])
