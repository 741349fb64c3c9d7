#!/usr/bin/env python3
"""Static instruction census of the macroblock kernel per source line, from a `hipcc -gline-tables-only -save-temps` .s:
   python tools/isa_lines.py file.s [kernel-substring] [file:lo-hi ...]
Prints instruction counts (SALU / VALU / LDS / VMEM / other) for the whole kernel and for the requested line ranges."""
import collections
import re
import sys

def main():
    path = sys.argv[1]
    kern = sys.argv[2] if len(sys.argv) > 2 else "h264e_mb_kernelILb1"
    ranges = []
    for a in sys.argv[3:]:
        f, r = a.split(":")
        lo, hi = r.split("-")
        ranges.append((f, int(lo), int(hi)))
    files = {}
    cur = None
    inside = False
    counts = collections.Counter()
    kinds = collections.defaultdict(collections.Counter)
    for line in open(path, errors="replace"):
        s = line.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s)
        if m:
            files[int(m.group(1))] = m.group(3).split("/")[-1]
            continue
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"', s)
        if m:
            files[int(m.group(1))] = m.group(2).split("/")[-1]
            continue
        if re.match(r'^[A-Za-z_.$][\w.$]*:', s):
            lab = s.split(":")[0]
            if lab.startswith("_Z") or lab.startswith("__"):
                inside = kern in lab
            continue
        if not inside:
            continue
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith(".") or s.startswith(";") or s.startswith("//"):
            continue
        op = s.split()[0]
        if op.startswith("s_"):
            k = "salu"
            if op.startswith("s_waitcnt") or op.startswith("s_nop"): k = "wait"
            elif op.startswith("s_cbranch") or op.startswith("s_branch"): k = "branch"
            elif op.startswith("s_load") or op.startswith("s_buffer"): k = "smem"
        elif op.startswith("v_"):
            k = "valu"
            if "readlane" in op or "writelane" in op or "readfirstlane" in op: k = "lane"
        elif op.startswith("ds_"):
            k = "lds"
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_") or op.startswith("flat_"):
            k = "vmem" if not op.startswith("scratch_") else "scratch"
        else:
            k = "other"
        counts[k] += 1
        if cur:
            kinds[cur][k] += 1
    print("kernel", kern, dict(counts), "total", sum(counts.values()))
    for f, lo, hi in ranges:
        c = collections.Counter()
        for (ff, ln), kk in kinds.items():
            if ff == f and lo <= ln <= hi:
                c.update(kk)
        print("%s:%d-%d" % (f, lo, hi), dict(c), "total", sum(c.values()))
    if not ranges:
        tot = [(sum(kk.values()), k, dict(kk)) for k, kk in kinds.items()]
        tot.sort(reverse=True)
        for n, k, d in tot[:60]:
            print(n, k, d)

if __name__ == "__main__":
    main()
