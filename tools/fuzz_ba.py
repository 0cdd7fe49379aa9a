"""Random local-BA / global-BA windows, GPU vs oracle at the parity bar of tests/test_gpu_ba.py (check_ba).  Window shape, observation
counts, outlier and monocular fractions, noise levels, gamma, iteration counts and the solver are drawn at random.
   python tools/fuzz_ba.py [n=200] [seed=0]"""
import sys, traceback, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from lld_slam_amd import Context, Optimizer, BABatch, synth
import oracle_py as O
from test_gpu_ba import check_ba
ctx = Context(0); O.lib()
import os
BIG = os.environ.get("FUZZ_BIG") == "1"


def rel(a, b): return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-3)


def deviation(a, o, w):
    return dict(cam=float(np.abs(a.cam_qt - o.cam_qt).max()) if w.n_cams else 0.0, pt=float(rel(a.pt_xyz, o.pt_xyz).max()) if w.n_points else 0.0,
                ln=float(rel(a.line_x0, o.line_x0).max()) if w.n_lines else 0.0,
                chi=abs(a.stats["chi2_final"] - o.stats["chi2_final"]) / max(o.stats["chi2_round1"], 1e-30))


def permuted(w, rng):
    """the same window with the landmarks in another order and every landmark's observations in another order - what the reference
    does from run to run (its lists and std::map<KeyFrame*, size_t> are in address order).  Returns (window, point order, line order):
    landmark i of the new window is landmark order[i] of the old one."""
    from lld_slam_amd import host
    def reorder(start, n_lm):
        order = rng.permutation(n_lm)
        new_start = [0]; idx = []
        for l_ in order:
            s_, e_ = start[l_], start[l_ + 1]
            idx.extend((s_ + rng.permutation(e_ - s_)).tolist()); new_start.append(len(idx))
        return order, np.array(new_start, np.int32), np.array(idx, np.int64)
    po, ps, pi = reorder(w.pt_obs_start, w.n_points)
    lo, ls, li = reorder(w.ln_obs_start, w.n_lines)
    w2 = host.Window(cam=w.cam, n_free_cams=w.n_free_cams, cam_qt=w.cam_qt.copy(), pt_xyz=w.pt_xyz[po], pt_obs_start=ps, pt_obs_cam=w.pt_obs_cam[pi],
                     pt_obs_uvr=w.pt_obs_uvr.reshape(-1, 3)[pi], pt_obs_inv_sigma2=w.pt_obs_inv_sigma2[pi], line_x0=w.line_x0[lo], line_dir=w.line_dir[lo],
                     ln_obs_start=ls, ln_obs_cam=w.ln_obs_cam[li], ln_obs_left=w.ln_obs_left.reshape(-1, 4)[li], ln_obs_right=w.ln_obs_right.reshape(-1, 4)[li],
                     ln_obs_octave=w.ln_obs_octave.reshape(-1, 2)[li])
    return w2.normalise(), po, lo


def deviation_permuted(b, po, lo, o, w):
    """b: result on the permuted window; back to the original landmark order"""
    import copy
    b2 = copy.copy(b)
    pt = np.empty_like(o.pt_xyz); pt[po] = b.pt_xyz; b2.pt_xyz = pt
    x0 = np.empty_like(o.line_x0); x0[lo] = b.line_x0; b2.line_x0 = x0
    return deviation(b2, o, w)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0; done = 0; soft = 0; by_tag = {}
batch = []
for it in range(n):
    n_free = int(rng.integers(0, 40)); n_fixed = int(rng.integers(1 if n_free == 0 else 0, 6))
    if BIG: n_free = int(rng.integers(20, 58)); n_fixed = int(rng.integers(0, 13))      # FUZZ_BIG=1: windows of the LBA-B class and beyond the 50-camera matrix-core limit
    if n_free + n_fixed < 2: n_fixed += 2
    kw = dict(n_free=n_free, n_fixed=n_fixed, n_points=int(rng.integers(0, 12000 if BIG else 900)), obs_per_point=int(rng.integers(2, min(12 if BIG else 7, n_free + n_fixed) + 1)),
              n_lines=int(rng.integers(0, 2500 if BIG else 150)), obs_per_line=int(rng.integers(1, min(8 if BIG else 6, n_free + n_fixed) + 1)), seed=int(rng.integers(1, 2 ** 31)),
              outlier_frac=float(rng.choice([0.0, 0.05, 0.2, 0.5])), mono_frac=float(rng.choice([0.0, 0.0, 0.3, 1.0])),
              mono_line_frac=float(rng.choice([0.0, 0.0, 0.4, 1.0])), noise=float(rng.choice([0.0, 0.5, 1.0, 3.0])),
              pose_sigma=(float(rng.uniform(0, 1.5)), float(rng.uniform(0, 0.15))), point_sigma=float(rng.uniform(0, 0.3)))
    par = dict(gamma=float(rng.choice([1.0, 1.0, 0.5, 0.1])), its_round1=int(rng.integers(1, 8)), its_round2=int(rng.integers(1, 18)))
    if rng.random() < 0.25: par = dict(protocol=1, its_round1=int(rng.integers(1, 12)), robust_points=int(rng.integers(0, 2)))   # Optimizer::BundleAdjustment
    if rng.random() < 0.2: par["abort_after_trials"] = int(rng.integers(1, 25))           # the stop flag raised after the k-th LM trial (round 3)
    try:
        w = synth.make_ba_window(**kw)
    except Exception as e:
        continue
    if w.n_edges() == 0 and rng.random() < 0.8: continue
    try:
        o = O.local_ba(w, **par)
        solver = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5])) if n_free <= 50 else 0      # round 5: 0 = structure-following (auto plan), 3 dense, 4 one chain, 5 two chains only
        g = Optimizer(ctx).LocalBundleAdjustment(w, reduced_solver=solver, **par)
        check_ba(g, o, w, tail="pcg" if (solver == 1 or n_free > 50) else None)      # the strict bar (every landmark to 1e-5, no twins: what exceeds it is classified below) unless the reduced solve is iterative
        done += 1
    except AssertionError as e:
        # beyond the bar: is it the reference's own order sensitivity on this window (an ill-conditioned one), or something else?
        same = (np.array_equal(g.pt_obs_outlier, o.pt_obs_outlier) and np.array_equal(g.ln_edge_outlier, o.ln_edge_outlier) and np.array_equal(g.line_removed, o.line_removed)
                and g.stats["lm_trials"] == o.stats["lm_trials"])
        dg = deviation(g, o, w)
        floor = dict(cam=0.0, pt=0.0, ln=0.0, chi=0.0)
        for r_ in range(6):
            w2, po, lo = permuted(w, rng)
            d2 = deviation_permuted(O.local_ba(w2, **par), po, lo, o, w)
            floor = {k_: max(floor[k_], d2[k_]) for k_ in floor}
        # ... or to HOW the landmark blocks are inverted / whether multiply-adds are fused (the oracle's rounding twins, round 3)
        from lld_slam_amd import host as _host
        try:
            O.set_landmark_inverse(1); d3 = deviation(O.local_ba(w, **par), o, w)
        finally:
            O.set_landmark_inverse(0)
        d4 = deviation(_host.ba_call(O.lib_fma(), None, w, _host.ba_params(O.lib_fma(), par.get("gamma", 1.0), **{k_: v_ for k_, v_ in par.items() if k_ != "gamma"})), o, w)
        floor = {k_: max(floor[k_], d3[k_], d4[k_]) for k_ in floor}
        bar = dict(cam=1e-7, pt=1e-5, ln=1e-5, chi=1e-5)                 # a quantity that is inside the bar needs no excuse
        within = all(dg[k_] <= max(bar[k_], 10 * floor[k_]) + 1e-12 for k_ in dg)
        tag = "FLOOR   "
        if not within:
            # round 5: ... or to the LAST BIT of the input (six copies with every pose / point component times 1 - 2^-52, 1 or 1 + 2^-52 at random): a window
            # whose LM run has two outcomes that far apart takes either (seed 99, window 6193: the oracle on such a copy lands on the device's result to 9 digits)
            import dataclasses
            for r_ in range(6):
                jig = lambda a: a * (1.0 + rng.integers(-1, 2, a.shape) * 2.0 ** -52)
                w3 = dataclasses.replace(w, cam_qt=jig(w.cam_qt), pt_xyz=jig(w.pt_xyz)) if r_ < 3 else dataclasses.replace(w, cam_qt=jig(w.cam_qt))
                d5 = deviation(O.local_ba(w3, **par), o, w)
                floor = {k_: max(floor[k_], d5[k_]) for k_ in floor}
            within = all(dg[k_] <= max(bar[k_], 10 * floor[k_]) + 1e-12 for k_ in dg)
            tag = "FLOOR/ulp"
        if not within and same and par.get("protocol", 0) == 0 and "abort_after_trials" not in par:
            # ... or the window has no isolated minimiser (noise-free observations and a free gauge or a camera left with too few inliers: chi2 -> 0 along a
            # valley).  Sign: ten more round-2 iterations lower BOTH costs tenfold and do not bring the two states closer (seed 99, window 892).
            p2 = dict(par, its_round2=par["its_round2"] + 10)
            o2 = O.local_ba(w, **p2); g2 = Optimizer(ctx).LocalBundleAdjustment(w, reduced_solver=solver, **p2)
            dg2 = deviation(g2, o2, w)
            if (o2.stats["chi2_final"] <= 0.1 * o.stats["chi2_final"] and g2.stats["chi2_final"] <= 0.1 * g.stats["chi2_final"] and max(dg2["cam"], dg2["pt"], dg2["ln"]) >= 0.5 * max(dg["cam"], dg["pt"], dg["ln"])):
                within = True; tag = "VALLEY  "
        if within: soft += 1; by_tag[tag.strip()] = by_tag.get(tag.strip(), 0) + 1
        else: bad += 1
        print(tag if within else "MISMATCH", it, "reduced_solver", solver, "trials gpu / oracle", g.stats["lm_trials"], o.stats["lm_trials"], "sets / trials equal", same, "gpu-oracle", {k_: "%.1e" % v_ for k_, v_ in dg.items()},
              "oracle vs its re-ordered / rounding twins", {k_: "%.1e" % v_ for k_, v_ in floor.items()}, kw if (not within or tag != "FLOOR   ") else "", par if (not within or tag != "FLOOR   ") else "", flush=True)
    except Exception as e:
        bad += 1; print("ERROR", it, kw, par, repr(e)[:300], flush=True)
# The excuses are counted one by one and bounded (a solver regression on ill-conditioned windows must not hide in them).  Bounds = about twice
# the rates of the runs on record (profiles/r0*_fuzz_ba_*.txt: FLOOR 2.3 - 2.7 %, FLOOR/ulp and VALLEY a handful per 20 000).
total = done + soft + bad
limits = {"FLOOR": 0.05, "FLOOR/ulp": 0.002, "VALLEY": 0.002}
over = {k_: v_ for k_, v_ in by_tag.items() if v_ > max(2, int(limits.get(k_, 0.0) * total))}
print("excused by kind:", {k_: by_tag.get(k_, 0) for k_ in limits}, "bounds", {k_: max(2, int(v_ * total)) for k_, v_ in limits.items()}, "exceeded:" if over else "none exceeded", over if over else "")
print("fuzzed", done + soft + bad, "windows:", done, "within the parity bar,", soft, "beyond it but within 10x the oracle's own sensitivity (re-ordered input, Cholesky-inverse and FMA twins; input moved by one unit in the last place) or on a window without an isolated minimiser (VALLEY),", bad, "mismatches / errors")
sys.exit(1 if (bad or over) else 0)
