"""Developer probe (CPU only): what ORDER could buy the headline launch. The restatement's iteration records give every instance's work (fixed part + KKT inverses + ADMM
iterations + residual evaluations, cycle weights from the phase timers); a list scheduler with 2048 slots then gives the makespan of any dispatch order: index order (what the
hardware does), random permutations, LPT with perfect knowledge, and two-phase schemes that run k iterations for everyone and order the rest by a feature of the state so far."""
import sys, numpy as np, heapq
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import binding as ob
from polympc_amd import workloads
B=4096; wl=workloads.robot_batch(B)
ss=ob.sqp_default_settings(); ss.max_iter=10; ss.line_search_max_iter=10
tr=np.zeros((B,10,8)); ob.bind_iteration_trace(ss,tr)
x,l,info=ob.sqp_solve_batch(ob.MODEL_ROBOT,6,1,0.0,2.0,B,wl["d"],wl["lbx"],wl["ubx"],sqp_settings=ss,pivot=ob.PIVOT_SWEEP,threads=8)
it=np.array([i.iter for i in info])
qpi=tr[:,:,5]   # qp iterations per SQP iteration
act=(np.arange(10)[None,:]<it[:,None])
rho_upd=(qpi>=50).astype(float)+(qpi>=100).astype(float)
work=act*(52e3+37e3*(1+rho_upd)+600*qpi+600*np.ceil(qpi/10)*6)   # cycles per iteration (phase profile of the shared kernel)
tot=work.sum(axis=1)/2400.0   # us at 2.4 GHz
print("life us: mean %.0f p50 %.0f p99 %.0f max %.0f; total/2048 = %.0f us"%(tot.mean(),np.percentile(tot,50),np.percentile(tot,99),tot.max(),tot.sum()/2048))
def sched(order,dur):
    h=[0.0]*2048; heapq.heapify(h)
    for b in order:
        t=heapq.heappop(h); heapq.heappush(h,t+dur[b])
    return max(h)
print("FIFO (index order): %.0f us"%sched(range(B),tot))
print("oracle LPT: %.0f us"%sched(np.argsort(-tot),tot))
# two-phase: iteration 1 for everyone (FIFO), then the rest ordered by a predictor from iteration 1
w1=work[:,0]/2400.0; rest=tot-w1
t1=sched(range(B),w1)
from scipy.stats import spearmanr
feats={"qp_iter_1":qpi[:,0],"primal_norm_1":tr[:,0,2],"dual_norm_1":tr[:,0,3],"viol_1":tr[:,0,7],"cost_1":tr[:,0,4],"x0 deviation":np.abs(wl["lbx"][:,18:21]-0.5).max(axis=1) if np.isfinite(wl["lbx"][:,18:21]).all() else np.zeros(B)}
for k,f in feats.items():
    rho=spearmanr(f,rest).correlation
    m=t1+sched(np.argsort(-f),rest)
    print("predictor %-14s rank corr %.2f  -> two-phase makespan %.0f us (phase 1 %.0f)"%(k,rho,m,t1))
print("two-phase with oracle LPT of the rest: %.0f us; with FIFO rest: %.0f"%(t1+sched(np.argsort(-rest),rest),t1+sched(range(B),rest)))
# after 2, 3 iterations
for kk in (2,3,4):
    wk=work[:,:kk].sum(axis=1)/2400.0; rk=tot-wk; tk=sched(range(B),wk)
    f=tr[:,kk-1,2]*1.0   # primal norm of iteration kk
    f2=qpi[:,:kk].sum(axis=1)
    for name,ff in (("primal_norm_%d"%kk,f),("qp_iters_sum_%d"%kk,f2),("viol_%d"%kk,tr[:,kk-1,7])):
        print("after %d iterations: predictor %-16s rank corr %.2f -> makespan %.0f us"%(kk,name,spearmanr(ff,rk).correlation,tk+sched(np.argsort(-ff),rk)))
    print("   oracle LPT of the rest after %d iterations: %.0f"%(kk,tk+sched(np.argsort(-rk),rk)))
rng=np.random.default_rng(0)
ms=[sched(rng.permutation(B),tot) for _ in range(200)]
print("random permutations: mean %.0f min %.0f max %.0f p10 %.0f p90 %.0f; index order %.0f"%(np.mean(ms),np.min(ms),np.max(ms),np.percentile(ms,10),np.percentile(ms,90),sched(range(B),tot)))
# which instances are long? position of the longest 100 in index order
long_idx=np.argsort(-tot)[:50]; print("indices of the 50 longest:",np.sort(long_idx)[-12:], "their durations", np.round(tot[np.sort(long_idx)[-6:]]))
# reversed order, interleaved orders
print("reversed %.0f  evens-then-odds %.0f  second-half-first %.0f"%(sched(range(B-1,-1,-1),tot),sched(list(range(0,B,2))+list(range(1,B,2)),tot),sched(list(range(2048,B))+list(range(2048)),tot)))
# (late round 6) sliced launches in index order: every pass runs at most k SQP iterations of every unfinished instance (the slice mechanism of the MPC step), no ordering knowledge
cw=np.cumsum(work,axis=1)/2400.0
def sliced(cuts,ovh=8.0):
    t=0.0; prev=np.zeros(B); lo=0
    for hi in cuts:
        w=(cw[:,hi-1]-(cw[:,lo-1] if lo else 0.0))
        live=np.nonzero(it>lo)[0]
        t+=sched(live,w)+ovh; lo=hi
    return t
for cuts in ((10,),(5,10),(6,10),(7,10),(4,7,10),(3,6,10),(3,5,7,10),(2,4,6,8,10),tuple(range(1,11))):
    print("passes ending at iterations %-22s -> %.0f us (8 us per relaunch)"%(cuts,sliced(cuts)))
print("iteration histogram:",np.bincount(it,minlength=11))
