#!/usr/bin/env python3
"""k_i4_sel (tables.h) from k_i4_lut (tables.h): the intra 4x4 predictions as selectors into the pool of border-derived samples.

k_i4_lut says, per mode and sample, which filter of which border samples E[a], E[b], E[c] predicts it (H.264 8.3.1.2); the kernel
(enc_kernels.h wave_i4_choose) gathers from a pool E | F3 | F2 | DC instead, so every lut entry has to be ONE pool element:
  type 0 (a + 2b + c + 2) >> 2 with a, b, c consecutive   -> F3[b]        (pool offset 16 + b)
  type 3 (a + 3b + 2) >> 2, which is F3 at the pool's end -> F3[b]        (E[-1] = E[0], E[13] = E[12])
  type 1 (a + b + 1) >> 1 with a, b consecutive           -> F2[min(a,b)] (pool offset 32 + ..)
  type 2 a                                                -> E[a]         (pool offset a)
  mode 2                                                  -> DC           (pool offset 47)
Run without arguments: regenerates the table, compares it with the one in tables.h and checks both against each other on random borders.
"""
import os, re, sys, random

HERE = os.path.dirname(os.path.abspath(__file__))
TABLES = os.path.join(HERE, "..", "h264-lab_amd", "csrc", "tables.h")


def parse(name, text):
    m = re.search(name + r"\[9\]\[\d+\]\s*=\s*\{(.*?)\};", text, re.S)
    rows = re.findall(r"\{([^{}]*)\}", m.group(1))
    return [[int(v, 0) for v in r.split(",") if v.strip()] for r in rows]


def selector(mode, entry):
    if mode == 2:
        return 47
    t, a, b, c = entry & 3, (entry >> 2) & 15, (entry >> 6) & 15, (entry >> 10) & 15
    if t == 2:
        return a
    if t == 1:
        assert abs(a - b) == 1, (mode, hex(entry))
        return 32 + min(a, b)
    if t == 0:
        assert abs(a - b) == 1 and abs(b - c) == 1 and a != c, (mode, hex(entry))
        return 16 + b
    # type 3: (a + 3b + 2) >> 2 = (b + 2b + a + 2) >> 2 = F3 at an end of the pool whose missing neighbour repeats the end sample
    assert (b, a) in ((0, 1), (12, 11)), (mode, hex(entry))
    return 16 + b


def generate(lut):
    return [[sum(selector(m, lut[m][4*y + x]) << (8*x) for x in range(4)) for y in range(4)] for m in range(9)]


def predict_lut(lut, mode, E, dc):
    out = []
    for e in lut[mode]:
        t, a, b, c = e & 3, (e >> 2) & 15, (e >> 6) & 15, (e >> 10) & 15
        out.append(dc if mode == 2 else [(E[a] + 2*E[b] + E[c] + 2) >> 2, (E[a] + E[b] + 1) >> 1, E[a], (E[a] + 3*E[b] + 2) >> 2][t])
    return out


def predict_sel(sel, mode, E, dc):
    pool = [0]*48
    for k in range(13):
        pool[k] = E[k]
        pool[16 + k] = (E[max(k - 1, 0)] + 2*E[k] + E[min(k + 1, 12)] + 2) >> 2
        if k < 12:
            pool[32 + k] = (E[k] + E[k + 1] + 1) >> 1
    pool[47] = dc
    return [pool[(sel[mode][y] >> (8*x)) & 255] for y in range(4) for x in range(4)]


def main():
    text = open(TABLES).read()
    lut, sel = parse("k_i4_lut", text), parse("k_i4_sel", text)
    gen = generate(lut)
    if "--print" in sys.argv:
        for r in gen:
            print("    { " + ", ".join("0x%08x" % v for v in r) + " },")
    assert gen == sel, "k_i4_sel in tables.h is not what k_i4_lut generates"
    rnd = random.Random(4)
    for _ in range(2000):
        E = [rnd.randrange(256) for _ in range(13)]
        dc = rnd.randrange(256)
        for m in range(9):
            assert predict_lut(lut, m, E, dc) == predict_sel(sel, m, E, dc), m
    print("k_i4_sel == generate(k_i4_lut); 2000 random borders x 9 modes agree")


if __name__ == "__main__":
    main()
