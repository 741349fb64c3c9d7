#!/usr/bin/env python3
"""How many macroblocks of a frame really depend on the rounded mv_clusters start candidates?  Encodes the first P frame of a GOP of
the synthetic clip twice, from two mv_clusters states whose rounded candidates differ by one full-pel step, and counts the macroblocks
whose records {mv[0], type} differ (study for DESIGN section 5: the validation treats every inter macroblock as a consumer)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _pkg  # noqa: E402

P = _pkg()
w, h = 1920, 1080


def mk(x, y):
    return ((y & 0xffff) << 16) | (x & 0xffff)


def first_p(t0, c1):
    ce = P.ClipEncoder(w, h, 2, gop=30, qp=26, clusters_in=(0, mk(*c1)), keep_records=1)
    ce.generate_synth(0, 2, t0=t0, seed=1)
    out, sizes, st = ce.encode()
    r = np.array(ce.read_records(1))
    ce.close()
    return r, sizes


for t0 in (16, 96, 288, 430, 442):
    for a, b in (((3, 1), (2, 1)), ((3, 3), (3, 2)), ((2, 3), (2, 2)), ((3, 3), (2, 2))):
        ra, sa = first_p(t0, a)
        rb, sb = first_p(t0, b)
        diff = np.nonzero((ra[:, :3] != rb[:, :3]).any(axis=1))[0]
        print("t0 %3d  c1 %s vs %s: %4d of %d macroblocks differ (first %s), frame bytes %d vs %d" %
              (t0, a, b, len(diff), len(ra), diff[:6].tolist(), sa[1], sb[1]), flush=True)
