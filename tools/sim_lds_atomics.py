"""Cost model of the LDS fp64 atomics of ba_linearize_pt on a synthetic LBA-B window (tools/microbench/lds_ops.hip gives the model: a
wavefront's ds_add_f64 is served in four groups of 16 consecutive lanes; a group costs, for its busiest 8-byte bank, 2 clocks per distinct
address and 3 per further lane on an address already counted).  Compares accumulator layouts / copy assignments.
   python tools/sim_lds_atomics.py"""
import sys, numpy as np
sys.path.insert(0, ".")
from lld_slam_amd import synth

def tasks_of(w):
    """lanes of each wavefront task: consecutive landmarks while their edges fit 64 lanes (host rule of lld_ba.hip)"""
    start = w.pt_obs_start; out = []; l = 0
    while l < w.n_points:
        e0 = start[l]; l1 = l
        while l1 < w.n_points and start[l1 + 1] - e0 <= 64: l1 += 1
        if l1 == l: l1 = l + 1
        out.append((e0, min(start[l1], e0 + 64))); l = l1
    return out

def group_cost(addr):
    """addr: addresses (in doubles) of the active lanes of one 16-lane group"""
    if len(addr) == 0: return 0
    banks = {}
    for a in addr: banks.setdefault(a % 32, []).append(a)
    c = 0
    for b, v in banks.items():
        d = len(set(v)); c = max(c, 2 * d + 3 * (len(v) - d))
    return max(c, 2)

def cost(w, addr_of):
    tot = 0; n = 0
    for e0, e1 in tasks_of(w):
        cams = w.pt_obs_cam[e0:e1]
        lanes = np.arange(e1 - e0)
        for g in range(4):
            sel = (lanes // 16 == g) & (cams < w.n_free_cams)
            tot += group_cost([addr_of(int(l), int(c)) for l, c in zip(lanes[sel], cams[sel])])
        n += 1
    return tot / n

w = synth.make_lba_b(0)
nf = w.n_free_cams
print("tasks", len(tasks_of(w)), "free cams", nf)
for copies, stride, name in [(4, 27, "now: 4 copies by lane>>3, stride 27"), (1, 27, "1 copy"), (2, 27, "2 copies by lane>>3"), (8, 27, "8 copies by lane>>3"),
                             (4, 28, "4 copies stride 28"), (4, 29, "4 copies stride 29"), (4, 31, "stride 31"), (4, 33, "stride 33")]:
    nacc = nf * stride
    print(f"{name:45s} {cost(w, lambda l, c: ((l >> 3) & (copies - 1)) * nacc + c * stride):6.2f} clocks per wavefront atomic")
for copies in (2, 4, 8):
    nacc = nf * 27
    print(f"{copies} copies by (lane % 16) // {16 // copies:<2d}".ljust(45), f"{cost(w, lambda l, c: ((l % 16) // (16 // copies)) * nacc + c * 27):6.2f}")
    print(f"{copies} copies by lane % {copies}".ljust(45), f"{cost(w, lambda l, c: (l % copies) * nacc + c * 27):6.2f}")
# copies by landmark slot: lanes of one landmark share a copy, neighbouring landmarks differ
def by_landmark(copies, pad):
    def f(l, c):
        return 0
    return f

def cost_greedy(w, copies, base_step, stride=27):
    """the host picks, lane by lane, the copy whose bank is least loaded in the lane's 16-lane group"""
    tot = 0; n = 0
    for e0, e1 in tasks_of(w):
        cams = w.pt_obs_cam[e0:e1]
        for g in range(4):
            load = {}; addrs = []
            for l in range(16 * g, min(16 * g + 16, e1 - e0)):
                c = int(cams[l])
                if c >= w.n_free_cams: continue
                best = None
                for q in range(copies):
                    a = (q, c); b = (q * base_step + c * stride) % 32
                    v = load.get(b, [])
                    d = len(set(v)); cur = 2 * d + 3 * (len(v) - d)
                    add = 3 if a in v else 2
                    if best is None or cur + add < best[0]: best = (cur + add, b, a)
                load.setdefault(best[1], []).append(best[2])
            c = 0
            for b, v in load.items():
                d = len(set(v)); c = max(c, 2 * d + 3 * (len(v) - d))
            tot += max(c, 2) if load else 0
        n += 1
    return tot / n
for copies, step in [(2, 16), (4, 8), (4, 1), (4, 3), (8, 4), (8, 1), (3, 11)]:
    print(f"host-chosen copy: {copies} copies, bases {step} banks apart".ljust(45), f"{cost_greedy(w, copies, step):6.2f}")
