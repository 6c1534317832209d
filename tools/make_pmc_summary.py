"""profiles/<tag>_pmc_summary.json from what tools/collect_profiles.sh left under gpurun_out/<tag>_prof (run here, after the GPU call):
step-kernel HBM traffic per launch per config (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 read correction of MI355X_MICROARCH.md
section HBM) and the reset path's SQ-counter issue rates; bench.py reads this file.  usage: python tools/make_pmc_summary.py r02"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", tag + "_prof")
out = {"_method": "rocprofv3 --pmc in separate passes with --kernel-trace only; per-launch medians.  traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024: "
                  "FETCH_SIZE reads half of a wide coalesced read stream on gfx950 (MI355X_MICROARCH.md, HBM); our 4-16 byte accesses are outside that "
                  "calibration, so the corrected figure is an upper bound and the uncorrected one is given beside it.",
       "step_kernel_traffic": {}, "step_kernel_traffic_uncorrected": {}, "step_kernel_write_bytes": {}}
hp = os.path.join(src, "source_hash.txt")
if os.path.exists(hp):  # the sources the measured library was built from, and when it was measured (tools/collect_profiles.sh)
    lines = open(hp).read().split()
    out["source_hash"], out["collected"] = lines[0], lines[1] if len(lines) > 1 else None
def step_of(summary):
    return next((v for k, v in summary.items() if "ge_k_step" in k), None)
for cfg in ("c2", "c3", "c4"):
    p = os.path.join(src, cfg, "summary.json")
    if not os.path.exists(p): continue
    sm = json.load(open(p)); st = step_of(sm)
    if st and "FETCH_SIZE" in st["counters"]:
        f, w = st["counters"]["FETCH_SIZE"], st["counters"]["WRITE_SIZE"]
        out["step_kernel_traffic"][cfg] = int((2 * f + w) * 1024); out["step_kernel_traffic_uncorrected"][cfg] = int((f + w) * 1024)
        out["step_kernel_write_bytes"][cfg] = int(w * 1024)
    if cfg == "c2":
        rp = {}
        for k, v in sm.items():
            if k.startswith("ge_k_features64") or k.startswith("ge_k_reset"):
                rp[k] = {x: v[x] for x in ("median_us", "valu_busy", "valu_insts_per_simd_cycle", "lds_busy", "lds_conflict_share", "avg_resident_waves_per_simd",
                                           "wave_share_valu", "wave_share_lds", "wave_share_wait_any", "wave_share_wait_inst_any", "clock_ghz", "kernel_cycles") if x in v}
                rp[k]["WRITE_SIZE_KB"] = v["counters"].get("WRITE_SIZE"); rp[k]["FETCH_SIZE_KB"] = v["counters"].get("FETCH_SIZE")
        rp["bound"] = "ge_k_features64: vector-ALU issue (valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles)); ge_k_reset: latency of the G(n,m) rejection loop (the slowest of ~2 700 slots)"
        out["reset_path"] = rp
    for f in ("summary.json", "kernel_stats.csv", "kernel_medians.txt", "gaps.txt"):
        if os.path.exists(os.path.join(src, cfg, f)): shutil.copy(os.path.join(src, cfg, f), os.path.join(ROOT, "profiles", f"{tag}_{cfg}_{f}"))
p = os.path.join(src, "step_1m_traffic.txt")
if os.path.exists(p):
    vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+ median=\s*([0-9.]+)", open(p).read())}
    if len(vals) == 2:
        out["step_kernel_traffic"]["c2_1m"] = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
        out["step_kernel_traffic_uncorrected"]["c2_1m"] = int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
        out["step_kernel_write_bytes"]["c2_1m"] = int(vals["WRITE_SIZE"] * 1024)
for f in ("bench_c2.json", "bench_c3.json", "bench_c4.json", "other_configs.jsonl", "source_hash.txt"):
    if os.path.exists(os.path.join(src, f)): shutil.copy(os.path.join(src, f), os.path.join(ROOT, "profiles", f"{tag}_{f}"))
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("step_kernel_traffic", "step_kernel_write_bytes")}))
