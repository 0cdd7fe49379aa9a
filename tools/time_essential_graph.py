"""Optimizer::OptimizeEssentialGraph through the C ABI: dense Cholesky on the matrix cores vs the matrix-free PCG.
Run on the GPU box: python tools/time_essential_graph.py [n_keyframes ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lld_slam_amd import Context, Optimizer, synth

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [300, 1000]
    opt = Optimizer(Context(0))
    for n in sizes:
        g = synth.make_essential_graph(0, n)
        for solver in (1, 2):
            if solver == 2 and n > 1200: continue
            opt.OptimizeEssentialGraph(g, solver=solver)
            t = time.perf_counter(); r = opt.OptimizeEssentialGraph(g, solver=solver); dt = time.perf_counter() - t
            print(f"{n} KF, {g.edge_i.shape[0]} edges, solver {solver}: {dt * 1e3:.1f} ms, chi2 {r.chi2:.6e}, LM {r.lm_iterations} it / {r.lm_trials} trials, PCG {r.pcg_iterations}")

if __name__ == "__main__":
    main()
