"""CPU model of the lean traversal's result-key structure (csrc/traverse_lean.hip.hpp: KeyWin + flat array + refill),
checked against a plain sorted multiset: same admission decisions, same `worst` after every admission.
Development aid for the window logic; the kernel itself is tested on the GPU against the oracle."""
import bisect
import random
import sys

TAKE = 48
INF = 0xFFFFFFFF


class Model:
    def __init__(self, ef):
        self.ef = ef
        self.res = 0
        self.low = []
        self.low_max = 0
        self.fill_max = 0
        self.wk, self.wc, self.T = [], [], INF   # ascending distinct keys + counts
        self.s_hint = 0
        self.refills = self.dense = self.spills = 0

    def worst(self):
        return self.wk[-1]

    def refill(self):
        assert not self.wk and self.low
        self.refills += 1
        kmax = self.low_max
        assert kmax == max(self.low), (kmax, max(self.low))
        s = self.s_hint
        while True:
            hist = [0] * 64
            for k in self.low:
                b = (kmax - k) >> s
                if b < 64:
                    hist[b] += 1
            if s == 0:
                tn = kmax - 63 if kmax >= 63 else 0
                break
            L = acc = 0
            while L < 64:
                if acc + hist[L] > TAKE:
                    break
                acc += hist[L]
                L += 1
            if L > 0:
                span = L << s
                tn = kmax + 1 - span if kmax + 1 > span else 0
                break
            s = s - 3 if s > 3 else 0
        taken = sorted(k for k in self.low if k >= tn)
        self.low = [k for k in self.low if k < tn]
        self.low_max = max(self.low) if self.low else 0
        if s == 0:
            self.dense += 1
            self.s_hint = 2 if len(taken) < 12 else 0
        else:
            assert len(taken) == acc <= TAKE
            self.s_hint = (min(s + 1, 10) if (L == 64 and acc < 24) else (s - 1 if L < 6 else s))
        self.wk, self.wc = [], []
        for k in taken:
            if self.wk and self.wk[-1] == k:
                self.wc[-1] += 1
            else:
                self.wk.append(k)
                self.wc.append(1)
        assert len(self.wk) <= 64
        self.T = tn

    def low_append(self, d):
        self.low.append(d)
        self.low_max = max(self.low_max, d)

    def win_insert(self, d):
        if d in self.wk:
            self.wc[self.wk.index(d)] += 1
            return
        assert len(self.wk) < 64  # entries >= T never grow in number; a dense window spans 64 key values
        r = sum(1 for k in self.wk if k < d)
        self.wk.insert(r, d)
        self.wc.insert(r, 1)

    def admit_list(self, ds):
        adm = []
        i = 0
        if self.res < self.ef:
            f = min(len(ds), self.ef - self.res)
            for d in ds[:f]:
                self.low.append(d)
                self.fill_max = max(self.fill_max, d)
                adm.append(True)
            self.res += f
            i = f
            if self.res >= self.ef:
                self.low_max = self.fill_max
                self.refill()
        for d in ds[i:]:
            if self.res >= self.ef and d < self.worst():
                adm.append(True)
                if self.wc[-1] > 1:
                    self.wc[-1] -= 1
                else:
                    self.wk.pop()
                    self.wc.pop()
                if d >= self.T:
                    self.win_insert(d)
                else:
                    self.low_append(d)
                if not self.wk:
                    self.refill()
            else:
                adm.append(False)
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
            tot = sum(m.wc) + len(m.low)
            assert tot == ef, (tot, ef)
            assert all(k < m.T for k in m.low) and all(k >= m.T for k in m.wk)
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
    }
    n = 0
    for name, g in gens.items():
        for ef in (1, 2, 10, 63, 64, 65, 100, 300, 1200, 2500):
            for seed in range(4):
                m = run(ef, g, 700, seed)
                n += 1
        print(name, "ok  (last: refills", m.refills, "dense", m.dense, "spills", m.spills, ")")
    print(n, "streams ok")


if __name__ == "__main__":
    main()
