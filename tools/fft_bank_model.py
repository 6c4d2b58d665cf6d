"""LDS bank-conflict model of fft_ola_wave_kernel for one plan pair (default 1176 -> 1280): per pass, the LDS-array
cycles of its data accesses by the rules of MI355X_MICROARCH.md (ds_read_b64: two groups of 32 lanes, 64 banks of
4 bytes; ds_write_b64: four groups of 16 lanes, 32 banks) against the conflict-free count.  Mirrors the index
expressions of fft_wave.hip (wave_first / wave_fused_first / wave_stage / post / filter passes); tables are left out
(their rows are read by constant index or with odd pitches).  usage: python tools/fft_bank_model.py"""
import sys


def read_cycles(addrs):
    """addrs: value (8-byte) addresses of the 64 lanes, None = masked."""
    total = 0
    for g in range(2):
        lanes = [a for a in addrs[32 * g:32 * g + 32] if a is not None]
        if not lanes:
            continue
        banks = {}
        for a in set(lanes):
            for d in (2 * a, 2 * a + 1):
                banks.setdefault(d % 64, set()).add(d)
        total += max(len(v) for v in banks.values())
    return total


def write_cycles(addrs):
    total = 0
    for g in range(4):
        lanes = [a for a in addrs[16 * g:16 * g + 16] if a is not None]
        if not lanes:
            continue
        banks = {}
        for a in set(lanes):
            for d in (2 * a, 2 * a + 1):
                banks.setdefault(d % 32, set()).add(d)
        total += max(len(v) for v in banks.values())
    return total


from math import gcd

NEW_RULES = True


class Plan:
    def __init__(self, n, radices):
        self.n, self.r = n, radices
        self.k = len(radices)
        self.fused = self.k >= 3 and radices[0] * radices[1] <= 21

    def unit(self):          # values a lane of the first pass writes side by side
        return self.r[0] * self.r[1] if self.fused else self.r[0]

    def next_stage(self):    # the stage that reads the first pass's output
        return 2 if self.fused else 1

    def padj(self):          # units of the first pass per padding value (0 = none)
        u, s = self.unit(), self.next_stage()
        if not NEW_RULES:
            return 8 if (self.fused and u % 2 == 0 and self.n // self.r[2] == 8 * u) else 0
        if u % 2 or s >= self.k:
            return 0
        p = 32 // gcd(2 * u, 32)
        return p if (self.n // self.r[s]) % (p * u) == 0 else 0

    def stride(self, s):
        v = 1
        for i in range(s):
            v *= self.r[i]
        return v

    def out_pad(self, s):
        if s < 1 or s + 1 >= self.k or (self.fused and s == 1):
            return 0
        if NEW_RULES:
            p = (16 - ((self.r[s] - 1) * self.stride(s)) % 16) % 16
            if self.stride(s) >= self.n // self.r[s]:   # one block: nothing to separate
                p = 0
        else:
            p = 2 if self.r[s] == 7 and self.stride(s) == 21 else 0
        return p if p and self.n // self.r[s + 1] == self.stride(s + 1) else 0

    def in_pad(self, s):
        if s == self.next_stage():
            pj = self.padj()
            return (self.n // self.r[s]) // (pj * self.unit()) if pj else 0
        return self.out_pad(s - 1) if s >= 2 else 0

    def in_period(self, s):   # elements of the stage's input between two padding values inside one q step (0 = none)
        if s == self.next_stage() and self.padj():
            per = self.padj() * self.unit()
            return per if self.n // self.r[s] > per else 0
        return 0

    def buf_values(self):
        pad = self.n // (self.padj() * self.unit()) if self.padj() else 0
        for s in range(1, self.k - 1):
            pad = max(pad, self.out_pad(s) * (self.n // self.stride(s + 1)))
        return self.n + 2 + pad


def model(plan, inverse, last_in_registers):
    rows = []
    n = plan.n
    if plan.fused:
        ra, rb = plan.r[0], plan.r[1]
        m2 = n // (ra * rb)
        padj = plan.padj()
        rd = wr = rd0 = wr0 = 0
        for it in range((m2 + 63) // 64):
            js = [l + 64 * it if l + 64 * it < m2 else None for l in range(64)]
            if inverse:   # inputs from LDS
                for m in range(ra * rb):
                    rd += read_cycles([None if j is None else j + m2 * m for j in js])
                    rd0 += 2 if any(j is not None for j in js[32:]) else 1
            for k in range(ra):
                for qq in range(rb):
                    wr += write_cycles([None if j is None else ra * rb * j + (j // padj if padj else 0) + k + ra * qq for j in js])
                    wr0 += sum(1 for g in range(4) if any(j is not None for j in js[16 * g:16 * g + 16]))
        rows.append(("fused first %dx%d" % (ra, rb), rd, rd0, wr, wr0))
        first_stage = 2
    else:
        r = plan.r[0]
        m = n // r
        padj = plan.padj()
        rd = wr = rd0 = wr0 = 0
        for it in range((m + 63) // 64):
            is_ = [l + 64 * it if l + 64 * it < m else None for l in range(64)]
            if inverse:
                for q in range(r):
                    rd += read_cycles([None if i is None else i + q * m for i in is_])
                    rd0 += 2 if any(i is not None for i in is_[32:]) else 1
            for q in range(r):
                wr += write_cycles([None if i is None else r * i + (i // padj if padj else 0) + q for i in is_])
                wr0 += sum(1 for g in range(4) if any(i is not None for i in is_[16 * g:16 * g + 16]))
        rows.append(("first %d" % r, rd, rd0, wr, wr0))
        first_stage = 1
    last = plan.k - 1
    for s in range(first_stage, plan.k):
        r, st = plan.r[s], plan.stride(s)
        m = n // r
        qs = m + plan.in_pad(s)
        ipp = plan.in_period(s)
        opad = plan.out_pad(s)
        rd = wr = rd0 = wr0 = 0
        for it in range((m + 63) // 64):
            is_ = [l + 64 * it if l + 64 * it < m else None for l in range(64)]
            for q in range(r):
                rd += read_cycles([None if i is None else i + (i // ipp if ipp else 0) + q * qs for i in is_])
                rd0 += 2 if any(i is not None for i in is_[32:]) else 1
            if s == last and last_in_registers:
                continue
            for q in range(r):
                wr += write_cycles([None if i is None else r * i - (r - 1) * (i % st) + opad * (i // st) + q * st for i in is_])
                wr0 += sum(1 for g in range(4) if any(i is not None for i in is_[16 * g:16 * g + 16]))
        rows.append(("stage %d (radix %d, stride %d)" % (s, r, st), rd, rd0, wr, wr0))
    return rows


def pair_pass(name, n2, iters, lo, hi):
    rd = wr = rd0 = wr0 = 0
    for trip in range((iters + 63) // 64):
        idx = [l + 64 * trip if l + 64 * trip < iters else None for l in range(64)]
        for f in (lo, hi):
            a = [None if i is None else f(i) for i in idx]
            rd += read_cycles(a)
            rd0 += 2 if any(i is not None for i in idx[32:]) else 1
            wr += write_cycles(a)
            wr0 += sum(1 for g in range(4) if any(i is not None for i in idx[16 * g:16 * g + 16]))
    return (name, rd, rd0, wr, wr0)


PLANS = {64: [8, 8], 128: [2, 8, 8], 256: [4, 8, 8], 512: [8, 8, 8], 768: [3, 4, 8, 8], 1024: [2, 8, 8, 8], 1536: [3, 8, 8, 8],
         2048: [4, 8, 8, 8], 588: [3, 4, 7, 7], 640: [2, 5, 8, 8], 882: [2, 3, 3, 7, 7], 1176: [3, 7, 7, 8], 1280: [4, 5, 8, 8],
         1764: [3, 3, 4, 7, 7], 2352: [2, 3, 7, 7, 8], 2560: [5, 8, 8, 8]}


def totals(fwd, inv):
    rows = model(fwd, False, False)
    rows.append(pair_pass("real-FFT post-process", fwd.n, fwd.n // 2 - 1, lambda i: 1 + i, lambda i: fwd.n - 1 - i))
    rows.append(pair_pass("filter + inverse pre-process", inv.n, (inv.n + 1) // 2 - 1, lambda i: 1 + i, lambda i: inv.n - 1 - i))
    rows += model(inv, True, inv.r[-1] % 2 == 0)
    return rows


def main():
    global NEW_RULES
    if len(sys.argv) > 1 and sys.argv[1] == "--all":
        for n, r in sorted(PLANS.items()):
            out = []
            for rules in (False, True):
                NEW_RULES = rules
                pl = Plan(n, r)
                f = model(pl, False, False)
                i = model(pl, True, False)
                out.append((sum(x[1] for x in i), sum(x[2] for x in i), sum(x[3] for x in f), sum(x[4] for x in f), pl.buf_values()))
            print("%5d %-16s old: reads %4d/%4d writes %4d/%4d buf %5d | new: reads %4d/%4d writes %4d/%4d buf %5d" %
                  (n, r, *out[0], *out[1]))
        return
    fwd, inv = Plan(1176, PLANS[1176]), Plan(1280, PLANS[1280])
    if len(sys.argv) > 2:
        fwd, inv = Plan(int(sys.argv[1]), PLANS[int(sys.argv[1])]), Plan(int(sys.argv[2]), PLANS[int(sys.argv[2])])
    for rules in (False, True):
        NEW_RULES = rules
        rows = totals(fwd, inv)
        tr = tr0 = tw = tw0 = 0
        print("rules:", "new" if rules else "old", " buffer values:", max(fwd.buf_values(), inv.buf_values()))
        print("%-36s %16s %16s" % ("pass", "read cycles", "write groups"))
        for name, rd, rd0, wr, wr0 in rows:
            print("%-36s %7d (min %4d) %7d (min %4d)" % (name, rd, rd0, wr, wr0))
            tr += rd; tr0 += rd0; tw += wr; tw0 += wr0
        print("%-36s %7d (min %4d) %7d (min %4d)" % ("total", tr, tr0, tw, tw0))


if __name__ == "__main__":
    main()
