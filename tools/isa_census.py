#!/usr/bin/env python3
"""Static per-phase instruction census of one macroblock-kernel variant (no GPU needed).

    python tools/isa_census.py [--kernel ILi1ELi2ELi4E] [--asm file.s] [--sub] > profiles/rNN_isa_census.txt

Compiles h264e_kernels.hip with `-gline-tables-only -save-temps` (unless --asm names an existing .s), then attributes every
instruction of the chosen kernel to a PHASE = the innermost function of the phase list it was inlined through (the `.loc` comments
carry the inline chain), and -- with --sub -- to the innermost helper of a second list inside that phase.  Function line ranges are
read from the sources, so the table follows the code.  Static counts: loops count once, both sides of a branch count; what the
table shows is where the CODE is and how scalar / vector it is, and how a change moves it -- the dynamic counts come from
rocprofv3 (profiles/rNN_sq_counters.json)."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "h264-lab_amd", "csrc")
PHASES = ["row_begin", "row_end", "finalize_frame", "export_frame", "device_clusters_walk", "load_top", "load_input", "wave_load_window", "window_advance", "row_prefetch",
          "diamond_g", "search_type", "search_8x8_wave", "inter_choose", "intra16_cost", "intra4_choose", "intra_merge", "mb_decide", "mb_write", "df_strength",
          "predict_chroma_inter", "wave_pred_chroma", "mb_recon_front", "mb_recon_back", "mb_intra_decide", "mb_search", "mb_ctx_init", "rowtask_load", "poll_progress", "lds_wait"]
SUBS = ["wave_xform_quant", "xform_quant_recon", "wave_recon", "cavlc_block", "cavlc_block_v", "bw_put", "bw_ue", "quant_luma_dc", "quant_chroma_dc", "halfpel3_win", "interp_core", "interp4_win", "interp4_hbm",
        "wave_i4_choose", "i4_block_code", "wave_deblock", "rv_wait_rect", "rv_wait_rect_g", "ref_load4", "mv_cost", "mvp_get_arr", "mvp_put_arr", "set_range",
        "grp_sad_ref", "wave_sad_ref_q", "wave_sad_ref", "wave_interp_chroma", "wave_interp_luma", "grp_interp_luma", "wave_copy_wh", "wave_pred16", "skip_chroma_ok",
        "partition_hints", "v16_fwd4x4", "v16_inv4x4"]


def function_ranges():
    """{file: [(lo, hi, name)]} for every function definition that starts in column 0 of the device headers"""
    out = {}
    for fn in os.listdir(CSRC):
        if not (fn.endswith(".h") or fn.endswith(".hip")):
            continue
        lines = open(os.path.join(CSRC, fn), errors="replace").read().split("\n")
        starts = []
        for i, l in enumerate(lines):
            m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:DEV|NOINLINE_DEV|DEVM|static|__global__)\b[^;{]*?\b([A-Za-z_]\w*)\s*\(", l)
            if m and not l.rstrip().endswith(";"):
                starts.append((i + 1, m.group(1)))
        rs = []
        for k, (lo, name) in enumerate(starts):
            hi = starts[k + 1][0] - 1 if k + 1 < len(starts) else len(lines)
            rs.append((lo, hi, name))
        out[fn] = rs
    return out


def fn_of(ranges, f, ln):
    for lo, hi, name in ranges.get(f, ()):
        if lo <= ln <= hi:
            return name
    return None


def cls(op):
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep") or op.startswith("s_barrier"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "lane"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    if op.startswith("v_"): return "valu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="ILi1ELi2ELi4E", help="substring of the mangled kernel name (default: narrow window, two waves, 4 per SIMD = the stream variant)")
    ap.add_argument("--asm", default=None)
    ap.add_argument("--sub", action="store_true")
    ap.add_argument("--defs", default="", help="extra -D flags for the compile, space separated")
    a = ap.parse_args()
    ranges = function_ranges()
    tmp = None
    asm = a.asm
    if not asm or not os.path.exists(asm):
        tmp = tempfile.mkdtemp(dir="/tmp")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-gline-tables-only", "-save-temps", "-c", os.path.join(CSRC, "h264e_kernels.hip"),
                               "-I" + os.path.join(ROOT, "include"), "-o", "k.o"] + a.defs.split(), cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        made = os.path.join(tmp, "h264e_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
        if a.asm:
            os.replace(made, a.asm)
            asm = a.asm
        else:
            asm = made
    inside = False
    chain = []
    cnt = collections.defaultdict(collections.Counter)
    sub = collections.defaultdict(collections.Counter)
    spill = collections.Counter()
    for line in open(asm, errors="replace"):
        s = line.strip()
        if s.startswith("_Z") and ":" in s[:200] and not s.startswith("_ZN"):
            inside = s.startswith("_Z15h264e_mb_kernel") and a.kernel in s.split(":")[0]
            continue
        if s.startswith(".Lfunc_end"):
            inside = False
            continue
        if not inside:
            continue
        if s.startswith(".loc"):
            chain = [(m.group(1), int(m.group(2))) for m in re.finditer(r"csrc/([A-Za-z_0-9.]+):(\d+)", s)]
            continue
        if not s or s.startswith(".") or s.startswith(";") or s.endswith(":"):
            continue
        op = s.split()[0]
        c = cls(op)
        names = [fn_of(ranges, f, ln) for f, ln in chain]           # innermost first
        phase = next((n for n in names if n in PHASES), None) or "row loop / other"           # the innermost phase function of the inline chain
        cnt[phase][c] += 1
        if a.sub:
            sb = next((n for n in names if n in SUBS), None)
            if sb:
                sub[(phase, sb)][c] += 1
        if op in ("v_writelane_b32", "v_readlane_b32") and "; 4-byte Folded" in line:
            spill[phase] += 1
    cols = ["salu", "valu", "lane", "lds", "vmem", "smem", "scratch", "branch", "wait", "other"]
    tot = collections.Counter()
    print("# static instruction census of h264e_mb_kernel<%s> (tools/isa_census.py); columns = instruction classes, 'lane' = v_readlane / v_writelane / v_readfirstlane" % a.kernel)
    print("%-26s %7s " % ("phase", "total") + " ".join("%7s" % c for c in cols))
    for k, v in sorted(cnt.items(), key=lambda kv: -sum(kv[1].values())):
        print("%-26s %7d " % (k, sum(v.values())) + " ".join("%7d" % v[c] for c in cols))
        tot.update(v)
    print("%-26s %7d " % ("ALL", sum(tot.values())) + " ".join("%7d" % tot[c] for c in cols))
    if a.sub:
        print("\n# helpers inside the phases (innermost helper of the list)")
        for (p, sb), v in sorted(sub.items(), key=lambda kv: -sum(kv[1].values())):
            if sum(v.values()) >= 40:
                print("%-26s %-20s %7d " % (p, sb, sum(v.values())) + " ".join("%7d" % v[c] for c in cols))


if __name__ == "__main__":
    main()
