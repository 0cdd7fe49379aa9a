"""One scene of a tools/fuzz_matchers.py campaign again, with the details of a mismatch (local_points and line_stereo so far):
   [FUZZ_BIG=1] python tools/exp_fuzz_matcher_scene.py local_points|line_stereo <seed> <it> <sid>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import fuzz_matchers as FZ
from fuzz_matchers import S, OS, synth, expect_slots
from lld_slam_amd import Context


def line_stereo(ctx, rng, sid):
    from lld_slam_amd import TwoFrameLineMatcher
    O = FZ.O
    nl, nr = FZ.size(rng, 400), FZ.size(rng, 400)
    s = synth.make_stereo_lines(sid, nl, nr, related_frac=float(rng.uniform(0.2, 0.95)), pixel_noise=float(rng.choice([0.2, 0.4, 2.0])))
    tau, ml = float(rng.choice([2.0, 1.2])), float(rng.choice([20, 5, 60]))
    print("left", nl, "right", nr, "tau", tau, "min length", ml)
    tm = TwoFrameLineMatcher(ctx, tau, K=s["K"], b=s["b"], minLineLength=ml)
    for rep in range(3):
        m, d, gate = tm.MatchLines(s["desc_left"], s["desc_right"], lines=s["left"], other_lines=s["right"], octaves=s["left_octave"], other_octaves=s["right_octave"], want_gate=True)
        me, de, ge = O.line_match_stereo(s["K"], s["b"], tau, ml, s["left"], s["left_octave"], s["desc_left"], s["right"], s["right_octave"], s["desc_right"], want_gate=True)
        gd = np.argwhere(gate != ge); md = np.nonzero(m != me)[0]
        dd = np.nonzero((m == me) & (m >= 0) & (d != de))[0]
        print("run", rep, "gate entries differing", len(gd), gd[:5].tolist(), "matches differing", md.size, md[:8].tolist(), m[md[:8]].tolist(), me[md[:8]].tolist(), "distances differing", dd.size)
        for i in md[:3]:
            for j in {int(m[i]), int(me[i])} - {-1}:
                bd = O.match_l2f32(s["desc_left"][i:i + 1], s["desc_right"][j:j + 1])
                print("   left", i, "right", j, "gate dev / oracle", gate[i, j], ge[i, j], "L2", float(bd[1][0]), "device d", d[i], "oracle d", de[i])
            row = np.linalg.norm(s["desc_left"][i].astype(np.float64) - s["desc_right"].astype(np.float64), axis=1)
            cand = np.nonzero(ge[i])[0]; order = cand[np.argsort(row[cand])][:4]
            print("   oracle-gated candidates by distance:", order.tolist(), row[order].tolist())


def main():
    name, seed, it, sid = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    ctx = Context(0); FZ.O.lib()
    rng = np.random.default_rng([seed, it])
    if name == "line_stereo": return line_stereo(ctx, rng, sid)
    assert name == "local_points"
    F, T, mp, nm = FZ._map_scene(rng, sid)
    view = S.frame_view(T, synth.KITTI_CAM, F)
    th, nn = float(rng.choice([1.0, 3.0, 5.0])), float(rng.choice([0.8, 0.6]))
    print("keypoints", F.n, "map points", nm, "th", th, "nnratio", nn)
    for rep in range(3):
        out, fr = S.search_local_points(ctx.lib, ctx.handle, F, view, mp, mp["occupied"], th, nn)
        k, inv, uvr, lvl, vc = OS.is_in_frustum(view, mp)
        m = inv != 0
        print("run", rep, "in view", int(m.sum()), "in_view equal", np.array_equal(fr["in_view"], inv), "uvr", np.array_equal(fr["proj_uvr"][m], uvr[m]),
              "view_cos", np.array_equal(fr["view_cos"][m], vc[m]), "level", np.array_equal(fr["level"][m], lvl[m]))
        ne, slot = OS.search_by_projection_map(F, mp["desc"], inv, uvr[:, :2], uvr[:, 2], lvl, vc, mp["has_obs"], mp["occupied"], th, nn)
        got = expect_slots(out, mp["occupied"])
        bad = np.nonzero(got != slot)[0]
        print("   n_matches", out.n_matches, ne, "rounds", out.rounds, "slots differing", bad.size, bad[:10], got[bad[:10]], slot[bad[:10]])
        for kp in bad[:4]:
            qs = [q for q in (got[kp], slot[kp]) if 0 <= q < nm]
            for q in qs:
                print("      keypoint", kp, "query", q, "device match / dist", out.match[q], out.best_dist[q], "level", lvl[q], "uv", uvr[q], "octave of kp", F.octave[kp], "xy", F.xy[kp])


if __name__ == "__main__":
    main()
