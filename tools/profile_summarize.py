#!/usr/bin/env python3
"""Turn the raw output of tools/profile_rNN.sh (gpurun_out/prof_rNN/) into the committed summaries under profiles/:
   python tools/profile_summarize.py gpurun_out/prof_r03 r03
writes profiles/r03_bench_kernel_stats.csv, r03_pmc_traffic.json, r03_sq_counters.json, r03_bench_line.json (+ _under_trace)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NMB, FRAMES, GOP = 8160, 600, 30
READ_I, READ_P, WRITE = 384, 768, 384


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(ROOT, "profiles")
    cmd = "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras  (tools/profile_%s.sh)" % tag
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, "%s_bench_kernel_stats.csv" % tag))
    for name, out in (("bench_line.json", "%s_bench_line.json"), ("bench_line_under_trace.json", "%s_bench_line_under_trace.json")):
        p = os.path.join(src, name)
        if os.path.exists(p):
            line = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if line:
                open(os.path.join(dst, out % tag), "w").write(line[-1] + "\n")
    # the variant the launches ran (kernel trace), and what the stream counted itself (bench line taken under the trace)
    kname = "h264e_mb_kernel"
    for l in open(os.path.join(src, "kernel_stats.csv")):
        if "h264e_mb_kernel" in l:
            kname = l.split('"')[1].replace("void ", "").split("(")[0] + "  (GEOM 1 = narrow window, WAVES per macroblock row, waves per SIMD aimed at)"
            break
    processed = None
    p = os.path.join(src, "bench_line_under_trace.json")
    if os.path.exists(p):
        line = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if line:
            processed = json.loads(line[-1]).get("config", {}).get("processed_mb_per_step")
    c = json.load(open(os.path.join(src, "counters.json")))
    cn, ln = c["counters"], c["launches"]
    passes = 3                                   # warmup 1 + steps 2
    launches = ln["FETCH_SIZE"]
    rd = sum((READ_P if f % GOP else READ_I) for f in range(FRAMES))*NMB*passes/launches
    rw = rd + FRAMES*NMB*WRITE*passes/launches
    fetch_kb, write_kb = cn["FETCH_SIZE"]/ln["FETCH_SIZE"], cn["WRITE_SIZE"]/ln["WRITE_SIZE"]
    total = (fetch_kb + write_kb)*1024.0
    json.dump({
        "command": "rocprofv3 --pmc FETCH_SIZE (and, in a separate run, --pmc WRITE_SIZE) -- " + cmd,
        "kernel": kname,
        "counters": {"FETCH_SIZE": {"launches": ln["FETCH_SIZE"], "sum_kb": cn["FETCH_SIZE"], "kb_per_launch": fetch_kb},
                     "WRITE_SIZE": {"launches": ln["WRITE_SIZE"], "sum_kb": cn["WRITE_SIZE"], "kb_per_launch": write_kb}},
        "bytes_per_launch": total, "algorithmic_read_bytes_per_launch": rd, "algorithmic_read_write_bytes_per_launch": rw,
        "ratio_vs_algorithmic_read_write": total/rw,
        "processed_macroblocks_per_pass": processed,
        "ratio_per_processed_macroblock": (total/rw)*(FRAMES*NMB)/processed if processed else None,
        "note": "raw counter values (KB) per launch, FETCH + WRITE; on gfx950 FETCH_SIZE under-reports wide streaming reads by up to 2x (MI355X_MICROARCH.md) "
                "and is uncalibrated for the 4..8-byte accesses of this kernel, so the fetch part lies between the raw value and twice it. About half of the "
                "macroblocks processed in a pass belong to frames thrown away at a mis-speculation abort."}, open(os.path.join(dst, "%s_pmc_traffic.json" % tag), "w"), indent=1)
    kc = cn["GRBM_GUI_ACTIVE"]/8.0
    simds = 1024
    # macroblocks PROCESSED per pass (incl. frames thrown away at aborts) are not counted by the hardware: report per USEFUL macroblock too
    useful = FRAMES*NMB*passes
    sq = {"command": "rocprofv3 --pmc SQ_* (two groups, separate runs) -- " + cmd + "; sums over %d launches = %d passes of the 600-frame clip" % (launches, passes)}
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
              "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        sq[k] = cn[k]
    allinst = cn["SQ_INSTS_VALU"] + cn["SQ_INSTS_SALU"] + cn["SQ_INSTS_LDS"] + cn["SQ_INSTS_SMEM"] + cn["SQ_INSTS_VMEM_RD"] + cn["SQ_INSTS_VMEM_WR"]
    sq.update({
        "kernel_cycles": kc, "kernel_cycles_source": "GRBM_GUI_ACTIVE / 8 XCDs", "simds": simds,
        "valu_issue_frac": cn["SQ_INSTS_VALU"]*4/(simds*kc), "salu_issue_frac": cn["SQ_INSTS_SALU"]*4/(simds*kc), "all_insts_issue_frac": allinst*4/(simds*kc),
        "insts_per_useful_macroblock": {"note": "%d passes x %d macroblocks of the stream (work thrown away at aborts is in the numerator only)" % (passes, FRAMES*NMB),
                                        "salu": cn["SQ_INSTS_SALU"]/useful, "valu": cn["SQ_INSTS_VALU"]/useful, "lds": cn["SQ_INSTS_LDS"]/useful,
                                        "vmem": (cn["SQ_INSTS_VMEM_RD"] + cn["SQ_INSTS_VMEM_WR"])/useful},
        "insts_per_processed_macroblock": ({"note": "per macroblock the rows actually got through (the kernel's own counter, bench line: %d per pass)" % processed,
                                            "salu": cn["SQ_INSTS_SALU"]/(processed*passes), "valu": cn["SQ_INSTS_VALU"]/(processed*passes),
                                            "lds": cn["SQ_INSTS_LDS"]/(processed*passes)} if processed else None),
        "wave_cycle_shares": {"parked (SQ_WAIT_ANY)": cn["SQ_WAIT_ANY"]/cn["SQ_WAVE_CYCLES"], "issue stalls (SQ_WAIT_INST_ANY)": cn["SQ_WAIT_INST_ANY"]/cn["SQ_WAVE_CYCLES"],
                              "issuing (SQ_ACTIVE_INST_ANY)": cn["SQ_ACTIVE_INST_ANY"]/cn["SQ_WAVE_CYCLES"]},
        "definition": "x_issue_frac = instructions x 4 cycles / (1024 SIMDs x kernel cycles): a wave64 VALU instruction holds its SIMD for 4 cycles, the scalar unit serves "
                      "each SIMD once per 4 cycles (MI355X_MICROARCH.md); VALU and SALU of different waves issue side by side, so either fraction is bounded by 1."})
    json.dump(sq, open(os.path.join(dst, "%s_sq_counters.json" % tag), "w"), indent=1)
    print(json.dumps({k: sq[k] for k in ("valu_issue_frac", "salu_issue_frac", "all_insts_issue_frac", "insts_per_useful_macroblock", "wave_cycle_shares")}, indent=1))
    print("traffic ratio vs algorithmic read+write: %.2f" % (total/rw))


if __name__ == "__main__":
    main()
