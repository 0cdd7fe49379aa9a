import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lld_slam_amd import Context, Optimizer, synth
ctx = Context(0); opt = Optimizer(ctx)
for i in range(6):
    f = synth.make_pose_frame(i)
    g = opt.PoseOptimization(f, gamma=0.5)
    print(i, "iterations", g.lm_iterations, "trials", g.lm_trials, "inliers", g.n_inliers)
