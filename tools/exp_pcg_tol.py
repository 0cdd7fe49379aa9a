import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
import oracle_py as O
from lld_slam_amd import synth, Context, Optimizer, BABatch
ctx=Context(0); opt=Optimizer(ctx)
rel=lambda x,y: np.linalg.norm(x-y,axis=1)/np.maximum(np.linalg.norm(y,axis=1),1e-3)
for name,w in (("small0",synth.make_lba_small(0)),("A",synth.make_lba_a(0)),("B",synth.make_lba_b(0))):
    o=O.local_ba(w)
    for tol in (1e-12,):
        g=opt.LocalBundleAdjustment(w, pcg_rel_tol=tol)
        g2=opt.LocalBundleAdjustment(w, pcg_rel_tol=tol)
        print(name,tol,'chi2 rel %.2e'%(abs(g.stats['chi2_final']-o.stats['chi2_final'])/o.stats['chi2_final']),
          'cam %.2e'%np.abs(g.cam_qt-o.cam_qt).max(),'pt %.2e'%rel(g.pt_xyz,o.pt_xyz).max(),'ln %.2e'%rel(g.line_x0,o.line_x0).max(),
          'r2r cam %.2e'%np.abs(g.cam_qt-g2.cam_qt).max(), 'pcg its',g.stats['pcg_iterations'],'trials',g.stats['lm_trials'],o.stats['lm_trials'])
# timing
for solver in (0, 1):
  for name,w,n in (("B",synth.make_lba_b(0),1),("B",synth.make_lba_b(0),64)):
    o=O.local_ba(w)
    b=BABatch(ctx,[w]*n, reduced_solver=solver)
    b.solve(); t=time.time(); b.solve(); dt=time.time()-t
    g=b.download(0)
    print('solver',solver,name,n,'solve s %.4f'%dt,'phase ms',np.round(b.phase_ms(),2),'launches',b.kernel_stats(1)[0],
          'chi2 rel %.2e'%(abs(g.stats['chi2_final']-o.stats['chi2_final'])/o.stats['chi2_final']),'cam %.2e'%np.abs(g.cam_qt-o.cam_qt).max())
    b.close()
