"""Monte-Carlo model of the pool kernel's batch scheduling (csrc/spt_pool.hip): how full batches are and what a
lane-bounce costs for a pool of P slots per wave under different pull policies.  Class arrival shares are the measured
ones of Cornell-9 (lanes per class in gpurun_out/pool1.log): DIFF 74.1 %, GEN 14.7 %, REFR 11.2 %."""
import random
import sys

SHARE = {"DIFF": .741, "GEN": .147, "REFR": .112}
COST = {"D1": 440.0, "DIFF": 150.0, "REFR": 200.0, "GEN": 130.0}   # VALU issue slots per batch (phase present or not)


def arrive(counts, k):
    for _ in range(k):
        r = random.random()
        acc = 0.0
        for c, p in SHARE.items():
            acc += p
            if r < acc:
                counts[c] += 1
                break


def sim(P, policy, thr, iters=40000):
    counts = {"DIFF": 0, "GEN": P, "REFR": 0}
    lanes = cost = 0.0
    for _ in range(iters):
        order = sorted(counts, key=lambda c: -counts[c])
        if policy == "single":
            full = [c for c in ("REFR", "GEN", "DIFF") if counts[c] >= 64]
            pick = [full[0] if full else order[0]]
        else:
            pick, room = [], 64
            for c in order:
                take = min(room, counts[c])
                if take == 0:
                    continue
                if pick and take < thr:      # top-up only if enough lanes share the extra phase
                    continue
                pick.append(c)
                room -= take
        room, tot, it_cost = 64, 0, COST["D1"]
        for c in pick:
            take = min(room, counts[c])
            counts[c] -= take
            room -= take
            tot += take
            it_cost += COST[c]
        arrive(counts, tot)
        lanes += tot
        cost += it_cost
    return lanes / iters / 64, cost / lanes


if __name__ == "__main__":
    random.seed(1)
    for P in (96, 128, 160, 192, 256):
        f, c = sim(P, "single", 0)
        line = f"P={P:3d} single-class: fill {f:.3f} cost/lane-bounce {c:.2f}"
        for thr in (8, 16, 24, 32):
            f, c = sim(P, "topup", thr)
            line += f" | top-up>={thr}: {f:.3f} {c:.2f}"
        print(line)
