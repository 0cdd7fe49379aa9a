import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
import oracle_py as O
from lld_slam_amd import synth, Context, Optimizer, BABatch
ctx=Context(0); opt=Optimizer(ctx)
rel=lambda x,y: np.linalg.norm(x-y,axis=1)/np.maximum(np.linalg.norm(y,axis=1),1e-3)
for name,w in (("small0",synth.make_lba_small(0)),):
    o=O.local_ba(w)
    for tol in (1e-12,):
        g=opt.LocalBundleAdjustment(w, pcg_rel_tol=tol)
        g2=opt.LocalBundleAdjustment(w, pcg_rel_tol=tol)
        print(name,tol,'chi2 rel %.2e'%(abs(g.stats['chi2_final']-o.stats['chi2_final'])/o.stats['chi2_final']),
          'cam %.2e'%np.abs(g.cam_qt-o.cam_qt).max(),'pt %.2e'%rel(g.pt_xyz,o.pt_xyz).max(),'ln %.2e'%rel(g.line_x0,o.line_x0).max(),
          'r2r cam %.2e'%np.abs(g.cam_qt-g2.cam_qt).max(), 'pcg its',g.stats['pcg_iterations'],'trials',g.stats['lm_trials'],o.stats['lm_trials'])
# timing
import os
for groups in (1,2,3,4):
  os.environ["LLD_BA_GROUPS"]=str(groups)
  for name,w,n in (("B",synth.make_lba_b(0),64),("B",synth.make_lba_b(0),256)):
    o=O.local_ba(w)
    b=BABatch(ctx,[w]*n)
    b.solve(); t=time.time(); b.solve(); dt=time.time()-t
    g=b.download(n-1)
    print('groups',groups,name,n,'solve s %.4f'%dt,'win/s %.0f'%(n/dt),'phase ms',np.round(b.phase_ms(),2),
          'chi2 rel %.2e'%(abs(g.stats['chi2_final']-o.stats['chi2_final'])/o.stats['chi2_final']))
    b.close()
