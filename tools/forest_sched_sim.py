"""tools/forest_sched_sim.py -- processor-sharing model of the tile kernel's launch (256 CUs x 7 resident workgroups, a
workgroup's time proportional to its tile's work items from tools/forest_items.py): launch order as it is vs heaviest tiles
first.  Needs /tmp/ipt_forest_<scale>.npy written by tools/forest_items.py forest."""
import numpy as np, heapq, sys
def simulate(work, order, n_cu=256, slots=7, fixed=200.0):
    # processor sharing per CU: each CU has capacity 1 item-unit per time unit, shared equally among resident WGs
    work = work[order] + fixed
    n = len(work)
    cu_res = [dict() for _ in range(n_cu)]  # wg -> remaining
    nxt = 0; t = 0.0
    # initial fill
    free = [(0, c) for c in range(n_cu) for _ in range(slots)]
    # event-driven: compute next completion among CUs
    rem = [ {} for _ in range(n_cu)]
    for c in range(n_cu):
        for _ in range(slots):
            if nxt < n: rem[c][nxt] = work[nxt]; nxt += 1
    while True:
        # next finishing event
        best = None
        for c in range(n_cu):
            if rem[c]:
                k = len(rem[c]); m = min(rem[c].values())
                dt = m * k
                if best is None or dt < best[0]: best = (dt, c)
        if best is None: break
        dt, cb = best
        t += dt
        for c in range(n_cu):
            if rem[c]:
                k = len(rem[c]); dec = dt / k
                done = [w for w, r in rem[c].items() if r - dec <= 1e-9]
                for w in rem[c]: rem[c][w] -= dec
                for w in done:
                    del rem[c][w]
                    if nxt < n: rem[c][nxt] = work[nxt]; nxt += 1
    return t
for scale in ("0.25", "1.0"):
    ipt = np.load(f"/tmp/ipt_forest_{scale}.npy")
    views = 20 if scale == "0.25" else 3
    T = len(ipt)
    rng = np.random.default_rng(1)
    # view-major natural order
    w = np.concatenate([np.roll(ipt, rng.integers(T)) for _ in range(views)])
    nat = np.arange(len(w))
    lpt = np.argsort(-w, kind='stable')
    ideal = (w.sum() + 200.0 * len(w)) / 256
    for name, od in (("natural", nat), ("lpt", lpt)):
        t = simulate(w, od)
        print(f"scale {scale} views {views}: {name:8s} makespan {t:12.0f}  ideal {ideal:12.0f}  ratio {t/ideal:.2f}")
