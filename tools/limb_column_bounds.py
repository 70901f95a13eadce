#!/usr/bin/env python3
"""Worst-case value of every 64-bit column accumulator in csrc/limb30.h's products (mulL, sqrL, mul2L).

All limbs are < 2^30 (the top limb too: values are < 630 p < 2^390), the Montgomery digit m is < 2^30 and the limbs of
p are the real ones, so every addition into a column has an exact upper bound.  This script replays the schedules of
limb30.h on those bounds -- rows of products, the reduction rows, the carry-outs (high dword x 4 into the next column,
low 32 bits kept) and the final normalisation -- and reports the largest value any column reaches, which must stay below
2^64.  With --search it also looks for the smallest contiguous carry-out sets (how the sets in limb30.h were chosen).
Run by tests/test_limb30_host.py."""
import sys

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
M = (1 << 30) - 1
PL = [(P >> (30 * j)) & M for j in range(13)]
LIMIT = 1 << 64


class Cols:
    def __init__(self):
        self.col = [0] * 28
        self.worst = 0

    def add(self, k, v):
        self.col[k] += v
        self.worst = max(self.worst, self.col[k])

    def carry_hi(self, cols):                       # col_carry_hi
        for c in cols:
            self.add(c + 1, (self.col[c] >> 32) << 2)
            self.col[c] = min(self.col[c], (1 << 32) - 1)

    def reduction_row(self, i):
        for j in range(13):
            self.add(i + j, M * PL[j])
        self.add(i + 1, self.col[i] >> 30)
        self.col[i] = 0

    def take_high(self):                            # fpl_take_high
        carry = 0
        for k in range(13):
            v = self.col[13 + k] + carry
            self.worst = max(self.worst, v)
            carry = v >> 30


def mul(step, cols):
    x = Cols()
    for i in range(13):
        for j in range(13):
            x.add(i + j, M * M)
        x.reduction_row(i)
        if i == step:
            x.carry_hi(cols)
    x.take_high()
    return x.worst


def sqr(cols):
    x = Cols()
    for i in range(13):
        x.add(2 * i, M * M)
        for j in range(i + 1, 13):
            x.add(i + j, 2 * M * M)
    x.carry_hi(cols)
    for i in range(13):
        x.reduction_row(i)
    x.take_high()
    return x.worst


def mul2(row, first, second):
    x = Cols()
    for i in range(13):
        for j in range(13):
            x.add(i + j, M * M)
        for j in range(13):
            x.add(i + j, M * M)
        if i == row:
            x.carry_hi(first)
    x.carry_hi(second)
    for i in range(13):
        x.reduction_row(i)
    x.take_high()
    return x.worst


def mul6(mid, between, last, pairs=3):
    """mul6L (mul4L: pairs = 2): pairs of products; per pair the rows alternate as in mul2L with a carry-out after row 7; between pairs a wider one"""
    x = Cols()
    for pair in range(pairs):
        for i in range(13):
            for j in range(13):
                x.add(i + j, M * M)
            for j in range(13):
                x.add(i + j, M * M)
            if i == 7:
                x.carry_hi(mid)
        if pair < pairs - 1:
            x.carry_hi(between)
    x.carry_hi(last)
    for i in range(13):
        x.reduction_row(i)
    x.take_high()
    return x.worst


def ranges(maxn):
    yield []
    for n in range(1, maxn + 1):
        for lo in range(0, 26 - n):
            yield list(range(lo, lo + n))


def main():
    used = {
        "mulL  (carry 10..14 after row 7)": mul(7, range(10, 15)),
        "sqrL  (carry 10..14 before the reduction)": sqr(range(10, 15)),
        "mul2L (carry 6..18 after row 7 of both, 12 before the reduction)": mul2(7, range(6, 19), [12]),
        "mul6L (6..18 after row 7 of every pair, 2..22 between pairs, 12 before the reduction)": mul6(range(6, 19), range(2, 23), [12]),
        "mul4L (the first two pairs of mul6L's schedule)": mul6(range(6, 19), range(2, 23), [12], pairs=2),
        "no carry-out at all, mulL": mul(-1, []),
    }
    ok = True
    for name, w in used.items():
        print("%-88s worst column = %.9f x 2^64" % (name, w / LIMIT))
        if "no carry" not in name:
            ok &= w < LIMIT
    if "--search" in sys.argv:
        for r in ranges(9):
            if mul(7, r) < LIMIT:
                print("smallest set for mulL after row 7:", r)
                break
        for r in ranges(9):
            if sqr(r) < LIMIT:
                print("smallest set for sqrL:", r)
                break
        best = None
        for row in range(4, 10):
            for a in ranges(14):
                if best and len(a) >= best[0]:
                    break
                for b in ranges(6):
                    if best and len(a) + len(b) >= best[0]:
                        break
                    if mul2(row, a, b) < LIMIT:
                        best = (len(a) + len(b), row, a, b)
                        break
        print("smallest sets for mul2L: %d carry-outs, after row %d: %s, before the reduction: %s" % best)
    print("OK" if ok else "OVERFLOW")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
