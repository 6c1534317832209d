"""Per-kernel medians of the counters collected by tools/pmc_sq_passes.sh, plus derived issue-rate fractions -> JSON.
usage: python tools/pmc_sq_summary.py <dir holding p1..pN and trace>"""
import csv, glob, json, os, statistics, sys
root = sys.argv[1]
acc, dur = {}, {}
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "ge_k" not in k: continue
        key = (k, r["Counter_Name"], r["Dispatch_Id"])
        per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    for (k, c, _), v in per.items(): acc.setdefault(k, {}).setdefault(c, []).append(v)
# durations from the FIRST counter pass (its kernel trace): the passes may run a serialized variant of the command (PMC_EXTRA of
# tools/pmc_sq_passes.sh: the shards of a sharded engine one after the other), and a launch's cycles must be those of the run its
# counters come from; kernel_medians.txt beside this file is the trace of the command as given
for f in (glob.glob(os.path.join(root, "p1", "**", "*_kernel_trace.csv"), recursive=True) or glob.glob(os.path.join(root, "trace", "**", "*_kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "ge_k" in k: dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
SIMDS = 256 * 4
# Cycles of a dispatch = its duration in the kernel trace x the shader clock measured INSIDE a kernel under this load (s_memtime against
# s_memrealtime, tools/phase_stamps.py -> profiles/r03_phase_stamps.txt: 2.22-2.40 GHz).  GRBM_GUI_ACTIVE / 8 reads high on dispatches
# shorter than ~0.3 ms (MI355X_MICROARCH.md) -- round 2's summaries derived "clocks" of 2.5-4.7 GHz from it and understated valu_busy.
CLOCK_GHZ = float(os.environ.get("GE_SHADER_CLOCK_GHZ", "2.25"))
out = {"_method": "rocprofv3 --pmc, separate passes with --kernel-trace only (MI355X_MICROARCH.md: rocprofv3 PMC slots); per-launch medians. "
                  "SQ_*_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs. "
                  "kernel_cycles = median duration (kernel trace) x shader clock measured in-kernel (GE_SHADER_CLOCK_GHZ, default 2.25); "
                  "valu_busy = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * kernel_cycles); lds_busy likewise per 256 CUs; "
                  "wave_share_* = share of SQ_WAVE_CYCLES a wave spends issuing VALU / LDS / waiting."}
for k, cs in sorted(acc.items()):
    m = {c: statistics.median(v) for c, v in cs.items()}
    d = {"launches": max(len(v) for v in cs.values()), "counters": m}
    if k in dur: d["median_us"] = statistics.median(dur[k])
    cyc = d.get("median_us", 0) * CLOCK_GHZ * 1e3
    if cyc > 0:
        d["kernel_cycles"] = cyc
        d["clock_ghz"] = CLOCK_GHZ
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc > 0:
        for name, c in (("valu", "SQ_ACTIVE_INST_VALU"), ("lds", "SQ_ACTIVE_INST_LDS"), ("scalar", "SQ_ACTIVE_INST_SCA"), ("any_inst", "SQ_ACTIVE_INST_ANY"),
                        ("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("wait_inst_lds", "SQ_WAIT_INST_LDS")):
            if c in m: d["wave_share_" + name] = m[c] / wc
    if cyc > 0 and "SQ_ACTIVE_INST_VALU" in m: d["valu_busy"] = m["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * cyc)
    if cyc > 0 and "SQ_INSTS_VALU" in m: d["valu_insts_per_simd_cycle"] = m["SQ_INSTS_VALU"] / (SIMDS * cyc)
    if cyc > 0 and "SQ_LDS_IDX_ACTIVE" in m: d["lds_busy"] = m["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE", 0) > 0: d["lds_conflict_share"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    if cyc > 0 and "SQ_WAVE_CYCLES" in m: d["avg_resident_waves_per_simd"] = m["SQ_WAVE_CYCLES"] * 4 / (SIMDS * cyc)
    out[k] = d
print(json.dumps(out, indent=1))
