"""Dev tool: turn `rocprofv3 --pmc ... --output-format csv` counter dumps into the JSON summaries kept under profiles/.

usage: python tools/pmc_summary.py traffic <fetch_csv> <write_csv> <out.json>
       python tools/pmc_summary.py sq2 <sq_csv> <f64mix_csv> <out.json> [forwards=20]   (round 2: adds VALU issue fraction and the
                                                                                            f64 instruction mix -> flops per env step)
       python tools/pmc_summary.py sq <sq_csv> <out.json> [forwards_per_launch_per_wave=20]
Only launches of sumo_step_kernel are used; the first `skip` launches (warm-up) are dropped."""
import csv, json, sys
from collections import defaultdict


KSUB = "sumo_step_kernel"       # overridden by a trailing `kernel=<substring>` argument; `steps=<K>`: env steps per env per launch
KSTEPS = 1
KENVS = 0
KSKIP = None                    # `skip=<n>`: leading launches of the kernel to drop (priming / warm-up)
KNOTE = ""                      # `note=<text>`: what was run (goes into the JSON's "method")
for _a in list(sys.argv):
    if _a.startswith("skip="):
        KSKIP = int(_a.split("=", 1)[1]); sys.argv.remove(_a)
    elif _a.startswith("note="):
        KNOTE = _a.split("=", 1)[1]; sys.argv.remove(_a)
for _a in list(sys.argv):
    if _a.startswith("envs="):
        KENVS = int(_a.split("=", 1)[1]); sys.argv.remove(_a)
for _a in list(sys.argv):
    if _a.startswith("kernel="):
        KSUB = _a.split("=", 1)[1]; sys.argv.remove(_a)
    elif _a.startswith("steps="):
        KSTEPS = int(_a.split("=", 1)[1]); sys.argv.remove(_a)


def per_launch(path, kernel_sub=None, skip=None):
    kernel_sub = kernel_sub or KSUB
    if skip is None:
        skip = KSKIP if KSKIP is not None else (5 if KSUB == "sumo_step_kernel" else 1)
    vals = defaultdict(lambda: defaultdict(float))    # counter -> dispatch -> value (summed over XCD/SE rows)
    name = None; grid = None
    for r in csv.DictReader(open(path)):
        if kernel_sub not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", ""); grid = int(r["Grid_Size"])
        vals[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {}
    for cn, d in vals.items():
        ids = sorted(d)[skip:]
        out[cn] = (sum(d[i] for i in ids) / max(1, len(ids)), len(ids))
    return name, grid, out


if sys.argv[1] == "traffic":
    name, grid, f = per_launch(sys.argv[2]); _, _, w = per_launch(sys.argv[3])
    fk, nf = f["FETCH_SIZE"]; wk, nw = w["WRITE_SIZE"]
    wgs = grid // 64
    # launches of <= 4096 envs carry ceil(N / 64) extra workgroups that rank the next launch's schedule (sumo_step): N + ceil(N/64) = wgs
    envs = wgs
    for n in range(wgs, 0, -1):
        if n <= 4096 and n + (n + 63) // 64 == wgs:
            envs = n
            break
    envs = KENVS or envs
    res = {"kernel": name, "envs": envs, "env_steps_per_env_per_launch": KSTEPS, "workgroups": wgs, "launches_averaged": [nf, nw],
           "traffic_bytes_per_env_step": (fk * 1024 * 2 + wk * 1024) / (envs * KSTEPS),
           "FETCH_SIZE_kb_per_launch": fk, "WRITE_SIZE_kb_per_launch": wk,
           "fetch_bytes_corrected_x2": fk * 1024 * 2, "write_bytes": wk * 1024,
           "traffic_bytes_per_launch": fk * 1024 * 2 + wk * 1024,
           "method": "two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only; " + (KNOTE or "bench.py fused launches") +
                     "; bytes = counter * 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                     "reports half of a coalesced stream; calibrated there for 16 B/lane loads, this kernel issues 8 B/lane record "
                     "reads, so treat the read side as approximate)"}
    json.dump(res, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(res, indent=1))
elif sys.argv[1] == "sq2":
    name, grid, s = per_launch(sys.argv[2]); _, _, m = per_launch(sys.argv[3])
    fw = float(sys.argv[5]) if len(sys.argv) > 5 else 20.0
    wgs = grid // 64
    envs = wgs
    for n in range(wgs, 0, -1):
        if n <= 4096 and n + (n + 63) // 64 == wgs:
            envs = n
            break
    envs = (KENVS or envs) * KSTEPS          # env steps per launch: every per-"wave" figure below is per env step
    avg = {k: v[0] for k, v in s.items()}
    mix = {k: v[0] for k, v in m.items()}
    d = {}
    d["valu_insts_per_forward_per_wave"] = avg["SQ_INSTS_VALU"] / envs / fw
    d["salu_insts_per_forward_per_wave"] = avg["SQ_INSTS_SALU"] / envs / fw
    d["lds_insts_per_forward_per_wave"] = avg["SQ_INSTS_LDS"] / envs / fw
    d["wave_cycles_per_forward"] = avg["SQ_WAVE_CYCLES"] / envs / fw * 4            # SQ_WAVE_CYCLES counts quad-cycles (MI355X_MICROARCH.md)
    for k, nm in (("SQ_ACTIVE_INST_VALU", "active_valu_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_WAIT_ANY", "wait_any_frac")):
        if k in avg: d[nm] = avg[k] / avg["SQ_WAVE_CYCLES"]
    # kernel cycles: GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3
    kcyc = avg["GRBM_GUI_ACTIVE"] / 8.0
    d["kernel_cycles"] = kcyc
    d["valu_issue_frac"] = avg["SQ_INSTS_VALU"] * 4.0 / (256 * 4 * kcyc)               # 4 issue cycles per wave64 VALU op, 1024 SIMDs
    if "SQ_THREAD_CYCLES_VALU" in mix and "SQ_ACTIVE_INST_VALU" in mix and mix["SQ_ACTIVE_INST_VALU"] > 0:
        d["mean_active_lanes"] = mix["SQ_THREAD_CYCLES_VALU"] / mix["SQ_ACTIVE_INST_VALU"]   # both in the same (quad-cycle) unit: counter_defs.yaml ratio
    f64 = {k: mix.get("SQ_INSTS_VALU_%s_F64" % k, 0.0) for k in ("ADD", "MUL", "FMA", "TRANS")}
    d["f64_wave_insts_per_env_step"] = {k: v / envs for k, v in f64.items()}
    d["f64_flops_issued_per_env_step"] = (f64["ADD"] + f64["MUL"] + f64["TRANS"] + 2.0 * f64["FMA"]) * 64.0 / envs
    d["f64_share_of_valu_insts"] = sum(f64.values()) / mix["SQ_INSTS_VALU"] if mix.get("SQ_INSTS_VALU") else None
    # issue time with f64 arithmetic priced at 4 cycles per wave64 instruction and every other VALU op at 2 (MI355X_MICROARCH.md: f64 runs
    # at half rate), and the flops that land on ACTIVE lanes (issued flops x mean active lanes / 64)
    if mix.get("SQ_INSTS_VALU"):
        nf64 = sum(f64.values()) * (avg["SQ_INSTS_VALU"] / mix["SQ_INSTS_VALU"])          # rescaled to the sq pass's launch average
        d["valu_issue_frac_priced"] = (4.0 * nf64 + 2.0 * (avg["SQ_INSTS_VALU"] - nf64)) / (256 * 4 * kcyc)
    if "mean_active_lanes" in d:
        d["f64_flops_on_active_lanes_per_env_step"] = d["f64_flops_issued_per_env_step"] * d["mean_active_lanes"] / 64.0
    res = {"kernel": name, "envs": envs // KSTEPS, "env_steps_per_env_per_launch": KSTEPS, "workgroups": wgs,
           "counters_avg_per_launch": dict(avg, **{"mix_" + k: v for k, v in mix.items()}),
           "derived": d,
           "method": "two rocprofv3 --pmc passes with --kernel-trace only; " + (KNOTE or "one env group, one fused launch per K steps") +
                     "; per-launch averages after the skipped priming launches; valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x "
                     "GRBM_GUI_ACTIVE/8); valu_issue_frac_priced = (4 x f64 + 2 x other VALU instructions) / the same denominator"}
    json.dump(res, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(res, indent=1))
elif sys.argv[1] == "mfma":
    # usage: pmc_summary.py mfma <counter_csv> <kernel_trace_csv or -> <out.json> kernel=<substring> flops=<algorithmic flops per launch>
    FL = 0.0
    for _a in list(sys.argv):
        if _a.startswith("flops="):
            FL = float(_a.split("=", 1)[1]); sys.argv.remove(_a)
    name, grid, s = per_launch(sys.argv[2])
    avg = {k: v[0] for k, v in s.items()}
    kcyc = avg["GRBM_GUI_ACTIVE"] / 8.0                                  # summed over the 8 XCDs by rocprofv3
    d = {"kernel_cycles": kcyc}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        # counts cycles (not quad-cycles) in which a SIMD's matrix pipe is busy, summed over SIMDs
        d["mfma_busy_frac_of_simd_time"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * kcyc)
    if "SQ_BUSY_CYCLES" in avg:
        d["sq_busy_cycles_per_launch"] = avg["SQ_BUSY_CYCLES"]
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in avg:
        d["mfma_flops_counted_per_launch"] = avg["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0      # the counter's unit is 512 flops
        d["mfma_tflops_at_2p4ghz"] = d["mfma_flops_counted_per_launch"] / (kcyc / 2.4e9) / 1e12
        d["frac_of_f32_mfma_peak_157p3"] = d["mfma_tflops_at_2p4ghz"] / 157.3
        if FL:
            d["algorithmic_flops_per_launch"] = FL
            d["counted_over_algorithmic"] = d["mfma_flops_counted_per_launch"] / FL
    if "SQ_INSTS_MFMA" in avg:
        d["mfma_insts_per_launch"] = avg["SQ_INSTS_MFMA"]
    if "SQ_WAVE_CYCLES" in avg and "SQ_WAIT_ANY" in avg:
        d["wait_any_frac"] = avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"]
    res = {"kernel": name, "grid": grid, "launches_averaged": max(v[1] for v in s.values()), "counters_avg_per_launch": avg, "derived": d,
           "method": "rocprofv3 --pmc pass with --kernel-trace only over `python3 tools/prof_workload.py mfma` (plain launches, no HIP graph); "
                     "kernel time = GRBM_GUI_ACTIVE / 8 cycles at 2.4 GHz; MFMA flops = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512; busy fraction = "
                     "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles). " + KNOTE}
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res, indent=1))
else:
    name, grid, s = per_launch(sys.argv[2])
    fw = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
    waves = grid // 64
    avg = {k: v[0] for k, v in s.items()}
    d = {}
    if "SQ_INSTS_VALU" in avg: d["valu_insts_per_forward_per_wave"] = avg["SQ_INSTS_VALU"] / waves / fw
    if "SQ_INSTS_SALU" in avg: d["salu_insts_per_forward_per_wave"] = avg["SQ_INSTS_SALU"] / waves / fw
    if "SQ_INSTS_LDS" in avg: d["lds_insts_per_forward_per_wave"] = avg["SQ_INSTS_LDS"] / waves / fw
    if "SQ_WAVE_CYCLES" in avg:
        d["wave_cycles_per_forward"] = avg["SQ_WAVE_CYCLES"] / waves / fw * 4   # SQ_WAVE_CYCLES counts in units of 4 cycles
        for k, nm in (("SQ_ACTIVE_INST_VALU", "active_valu_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_WAIT_ANY", "wait_any_frac")):
            if k in avg: d[nm] = avg[k] / avg["SQ_WAVE_CYCLES"]
    res = {"kernel": name, "envs": waves, "counters_avg_per_launch": avg, "derived": d}
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res, indent=1))
