import sys; sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import numpy as np, oracle_py as O
from lld_slam_amd import synth, Context, Optimizer
def dev(a,b):
    dq=np.abs(a.sim3[:,:4]-b.sim3[:,:4]).max()
    dt=(np.linalg.norm(a.sim3[:,4:7]-b.sim3[:,4:7],axis=1)/np.maximum(1.0,np.linalg.norm(b.sim3[:,4:7],axis=1))).max()
    ds=np.abs(a.sim3[:,7]/b.sim3[:,7]-1).max()
    return ["%.1e"%x for x in (dq,dt,ds,abs(a.chi2-b.chi2)/max(b.chi2,1e-300))]
ctx=Context(0); opt=Optimizer(ctx)
for gid,n,fix in [(0,120,True),(2,60,False),(3,300,True),(6,7,True),(7,16,True),(8,17,False)]:
    gr=synth.make_essential_graph(gid,n)
    for k in (15,1,2,3):
        o=O.optimize_essential_graph(gr,bFixScale=fix,iterations=k)
        for solver in (1,2):
            g=opt.OptimizeEssentialGraph(gr,bFixScale=fix,solver=solver,iterations=k)
            print(gid,n,fix,"k",k,"solver",solver,"oracle it/tr",o.lm_iterations,o.lm_trials,"%.6e"%o.chi2,"gpu",g.lm_iterations,g.lm_trials,"%.6e"%g.chi2,dev(g,o),flush=True)
