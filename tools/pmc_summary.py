"""Dev tool: turn `rocprofv3 --pmc ... --output-format csv` counter dumps into the JSON summaries kept under profiles/.

usage: python tools/pmc_summary.py traffic <fetch_csv> <write_csv> <out.json>
       python tools/pmc_summary.py sq <sq_csv> <out.json> [forwards_per_launch_per_wave=20]
Only launches of sumo_step_kernel are used; the first `skip` launches (warm-up) are dropped."""
import csv, json, sys
from collections import defaultdict


def per_launch(path, kernel_sub="sumo_step_kernel", skip=5):
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
    res = {"kernel": name, "envs": envs, "workgroups": wgs, "launches_averaged": [nf, nw],
           "FETCH_SIZE_kb_per_launch": fk, "WRITE_SIZE_kb_per_launch": wk,
           "fetch_bytes_corrected_x2": fk * 1024 * 2, "write_bytes": wk * 1024,
           "traffic_bytes_per_launch": fk * 1024 * 2 + wk * 1024,
           "method": "two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, bench.py --steps 10 "
                     "--warmup 5 --no-cpu-baseline; bytes = counter * 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                     "reports half of a coalesced stream; calibrated there for 16 B/lane loads, this kernel issues 8 B/lane record "
                     "reads, so treat the read side as approximate)"}
    json.dump(res, open(sys.argv[4], "w"), indent=1)
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
