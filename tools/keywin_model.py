"""CPU model of the lean traversal's result-key structure (csrc/traverse_lean.hip.hpp: KeyWin + flat array + refill),
checked against a plain sorted multiset: same admission decisions, same `worst` after every list.
Development aid for the window logic; the kernel itself is tested on the GPU against the oracle."""
import bisect
import random

TAKE = 48
MAXIN = 12
SLACK = 192
INF = 1 << 16


class Model:
    def __init__(self, ef):
        self.ef = ef
        self.res = 0
        self.low = []            # flat array; entries >= T are dead (already absorbed once)
        self.wk, self.wc, self.T = [], [], INF   # ascending distinct keys + counts
        self.total = 0
        self.s_hint = 0
        self.refills = self.dense = self.slow = self.fast = self.compactions = 0

    def worst(self):
        return self.wk[-1]

    def alive(self):
        return [k for k in self.low if k < self.T]

    def refill(self):
        assert not self.wk
        self.refills += 1
        T = self.T
        s = self.s_hint
        while True:
            hist = [0] * 64
            for k in self.low:
                if k < T:
                    b = (T - 1 - k) >> s
                    if b < 64:
                        hist[b] += 1
            if sum(hist) == 0:          # nothing alive within 64 bins below T: the floor drops past the gap, bins get coarser
                span = 64 << s
                assert T > span, "refill with nothing alive"
                T -= span
                s = min(s + 3, 10)
                continue
            f = next(b for b in range(64) if hist[b])
            T -= f << s                 # no alive key in [T - (f << s), T): the floor may drop to the first occupied bin
            hist = hist[f:] + [0] * f
            limit = 64 - f              # bins counted below the new floor
            if s == 0:
                tn = T - limit if T >= limit else 0
                break
            L = acc = 0
            while L < limit:
                if acc + hist[L] > TAKE:
                    break
                acc += hist[L]
                L += 1
            if L == 0:                  # the top bin alone is too crowded: finer bins
                s = s - 3 if s > 3 else 0
                continue
            span = L << s
            tn = T - span if T > span else 0
            break
        taken = sorted(k for k in self.low if tn <= k < T)
        if s == 0:
            self.dense += 1
            n = len(taken)
            self.s_hint = 0 if n >= 24 else (1 if n >= 12 else (2 if n >= 6 else 3))
        else:
            assert len(taken) == acc <= TAKE
            if L == limit and acc < 24:
                self.s_hint = min(s + 1, 10)
            elif span <= 80:
                self.s_hint = 0
            elif L < 16:
                self.s_hint = s - 1
            else:
                self.s_hint = s
        self.wk, self.wc = [], []
        for k in taken:
            if self.wk and self.wk[-1] == k:
                self.wc[-1] += 1
            else:
                self.wk.append(k)
                self.wc.append(1)
        assert len(self.wk) <= 64
        self.total = len(taken)
        self.T = tn

    def low_append(self, ds):
        if len(self.low) + len(ds) > self.ef + SLACK:
            self.compactions += 1
            self.low = [k for k in self.low if k < self.T]
        self.low.extend(ds)
        assert len(self.low) <= self.ef + SLACK

    def win_insert(self, d):
        if d in self.wk:
            self.wc[self.wk.index(d)] += 1
        else:
            assert len(self.wk) < 64
            r = sum(1 for k in self.wk if k < d)
            self.wk.insert(r, d)
            self.wc.insert(r, 1)
        self.total += 1

    def remove_largest(self, m):
        self.total -= m
        while m > 0:
            c = self.wc[-1]
            if c > m:
                self.wc[-1] -= m
                m = 0
            else:
                m -= c
                self.wk.pop()
                self.wc.pop()

    def admit_list(self, ds):
        n = len(ds)
        adm = [False] * n
        i0 = 0
        if self.res < self.ef:
            f = min(n, self.ef - self.res)
            self.low.extend(ds[:f])
            for i in range(f):
                adm[i] = True
            self.res += f
            i0 = f
            if self.res >= self.ef:
                self.T = max(self.low) + 1
                self.refill()
        if i0 >= n or self.res < self.ef:
            return adm
        w0 = self.worst()
        B = [i for i in range(i0, n) if ds[i] < w0]
        if not B:
            return adm
        inwin = [i for i in B if ds[i] >= self.T]
        if len(B) <= self.total and len(inwin) <= MAXIN:
            self.fast += 1
            ok = [i for i in B if ds[i] < self.T]
            for i in inwin:
                C = sum(1 for j in B if j < i and ds[j] <= ds[i])
                le = sum(c for k, c in zip(self.wk, self.wc) if k <= ds[i])
                if C < self.total - le:
                    ok.append(i)
            for i in ok:
                adm[i] = True
            for i in sorted(ok):
                if ds[i] >= self.T:
                    self.win_insert(ds[i])
            self.low_append([ds[i] for i in sorted(ok) if ds[i] < self.T])
            self.remove_largest(len(ok))
            if not self.wk:
                self.refill()
        else:
            self.slow += 1
            for i in range(i0, n):
                d = ds[i]
                if d < self.worst():
                    adm[i] = True
                    self.remove_largest(1)
                    if d >= self.T:
                        self.win_insert(d)
                    else:
                        self.low_append([d])
                    if not self.wk:
                        self.refill()
        return adm


class Ref:
    def __init__(self, ef):
        self.ef, self.s = ef, []

    def admit_list(self, ds):
        out = []
        for d in ds:
            if len(self.s) < self.ef:
                bisect.insort(self.s, d)
                out.append(True)
            elif d < self.s[-1]:
                self.s.pop()
                bisect.insort(self.s, d)
                out.append(True)
            else:
                out.append(False)
        return out


def run(ef, gen, lists, seed):
    rng = random.Random(seed)
    m, r = Model(ef), Ref(ef)
    for li in range(lists):
        ds = gen(rng, li)
        a, b = m.admit_list(ds), r.admit_list(ds)
        assert a == b, (ef, seed, li, ds, a, b)
        if m.res >= ef:
            assert m.worst() == r.s[-1], (ef, seed, li)
            assert m.total == sum(m.wc)
            assert m.total + len(m.alive()) == ef, (m.total, len(m.alive()), ef)
            assert all(k >= m.T for k in m.wk)
    return m


def main():
    gens = {
        "dense": lambda rng, li: [max(0, int(rng.gauss(32000 - 5 * li, 1600))) for _ in range(rng.randint(0, 32))],
        "sparse": lambda rng, li: [rng.randint(0, 65535) for _ in range(rng.randint(0, 8))],
        "ties": lambda rng, li: [900 * rng.randint(0, 40) for _ in range(rng.randint(0, 32))],
        "falling": lambda rng, li: [max(0, 60000 - 20 * li - rng.randint(0, 300)) for _ in range(rng.randint(1, 64))],
        "tiny": lambda rng, li: [rng.randint(0, 50) for _ in range(rng.randint(0, 64))],
        "same": lambda rng, li: [1234 for _ in range(rng.randint(0, 20))],
        "mixed": lambda rng, li: [rng.choice([rng.randint(0, 65535), 30000 + rng.randint(0, 40)]) for _ in range(rng.randint(0, 40))],
        "near-top": lambda rng, li: [max(0, 40000 - 3 * li - rng.randint(0, 25)) for _ in range(rng.randint(0, 64))],
    }
    n = 0
    for name, g in gens.items():
        for ef in (1, 2, 10, 63, 64, 65, 100, 300, 1200, 2500):
            for seed in range(4):
                m = run(ef, g, 700, seed)
                n += 1
        print(name, "ok  (last: refills", m.refills, "dense", m.dense, "fast", m.fast, "slow", m.slow, "compactions", m.compactions, ")")
    print(n, "streams ok")


if __name__ == "__main__":
    main()
