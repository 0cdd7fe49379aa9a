"""Random essential graphs for Optimizer::OptimizeEssentialGraph, GPU vs oracle at the bar of tests/test_gpu_posegraph.py (_check): rotation /
translation / scale / chi2 within 1e-5 or ten times the oracle's own FMA-twin distance.  A graph beyond that is compared with a wider
sample of the oracle's own spread - the FMA twin, four re-ordered edge lists and eight copies of the input moved by one unit in the last
place (every vertex and measurement component times 1 - 2^-52, 1 or 1 + 2^-52 at random), each with and without FMA.  One twin is one
draw of a heavy-tailed quantity: g2o differentiates EdgeSim3 numerically, and in the flat valley of a free-scale graph LM's accept /
reject decisions hinge on chi2 differences at rounding level; re-ordering the edges turned out not to move the oracle at all (its
normal equations are summed per block in a fixed order), the last-place jitter does.  FLOOR = within ten times that spread, iteration /
trial counts inside the range the variants span (+- 2 / 12 as in the test).   python tools/fuzz_posegraph.py [n=200] [seed=0]"""
import dataclasses, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from lld_slam_amd import Context, Optimizer, synth
import oracle_py as O
from test_gpu_posegraph import _check, deviation


def oracle_spread(gr, o, fix, iters, rng):
    spread = np.zeros(4); its = [o.lm_iterations, o.lm_iterations]; trials = [o.lm_trials, o.lm_trials]
    for k in range(13):
        perm = np.arange(gr.edge_i.size) if k == 0 or k > 4 else rng.permutation(gr.edge_i.size)
        q = dataclasses.replace(gr, edge_i=gr.edge_i[perm], edge_j=gr.edge_j[perm], edge_sji=gr.edge_sji[perm])
        if k > 4:                                                      # one unit in the last place, at random
            ulp = lambda a: a * (1.0 + rng.integers(-1, 2, a.shape) * 2.0 ** -52)
            q = dataclasses.replace(q, sim3=ulp(gr.sim3), edge_sji=ulp(gr.edge_sji))
        for fma in ((True,) if k == 0 else (False, True)):
            v = O.optimize_essential_graph(q, bFixScale=fix, iterations=iters, fma=fma)
            spread = np.maximum(spread, deviation(v, o))
            its = [min(its[0], v.lm_iterations), max(its[1], v.lm_iterations)]; trials = [min(trials[0], v.lm_trials), max(trials[1], v.lm_trials)]
    return spread, its, trials


def main():
    ctx = Context(0); O.lib()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = done = plain = at_floor = 0
    for it in range(n):
        nk = int(rng.choice([5, 7, 12, 16, 30, 60, 120]))
        kw = dict(drift=(float(rng.choice([0.0005, 0.002, 0.01])), float(rng.choice([0.01, 0.03, 0.1]))), covis=int(rng.choice([1, 3, 6])),
                  n_corrected=int(rng.integers(1, max(2, nk // 3))))
        fix = bool(rng.integers(0, 2)); solver = int(rng.choice([1, 2])); iters = int(rng.choice([15, 15, 2, 3]))
        gr = synth.make_essential_graph(int(rng.integers(0, 1 << 30)), nk, **kw)
        try:
            g = Optimizer(ctx).OptimizeEssentialGraph(gr, bFixScale=fix, solver=solver, iterations=iters)
            o = O.optimize_essential_graph(gr, bFixScale=fix, iterations=iters)
            floor = deviation(O.optimize_essential_graph(gr, bFixScale=fix, iterations=iters, fma=True), o)
            d = deviation(g, o)
            _check(g, o, floor); done += 1; plain += int((d <= 1e-5).all())
        except AssertionError:
            spread, its, trials = oracle_spread(gr, o, fix, iters, np.random.default_rng(it))
            ok = (d <= np.maximum(1e-5, 10 * spread)).all() and its[0] - 2 <= g.lm_iterations <= its[1] + 2 and trials[0] - 12 <= g.lm_trials <= trials[1] + 12
            if ok: at_floor += 1
            else: bad += 1
            print("FLOOR   " if ok else "MISMATCH", it, "n_kf", nk, kw, "fix", fix, "solver", solver, "iterations", iters, "deviation", d, "floor", floor, "its", g.lm_iterations, o.lm_iterations,
                  "trials", g.lm_trials, o.lm_trials, "oracle variants: spread", spread, "its", its, "trials", trials, flush=True)
        except Exception as e:
            bad += 1; print("ERROR", it, nk, kw, repr(e)[:200], flush=True)
    print("fuzzed", done + at_floor + bad, "graphs:", done, "within the bar (", plain, "of them within the plain 1e-5 ),", at_floor,
          "at the oracle's own floor (ten times the spread of its FMA twin, four re-ordered edge lists and eight inputs moved by one unit in the last place),", bad, "mismatches / errors")


if __name__ == "__main__":
    main()
