#!/bin/bash
# development: how busy the vector-memory path (TA / TCP) and the LDS are during the fused rollout launch.  One small counter set per pass
# (a set the hardware cannot collect makes rocprofv3 abort and hang in its finaliser: every pass runs under its own timeout).
export TMPDIR=/tmp
O=gpurun_out/qpmc3; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 20 --state-warmup 40 --no-cpu-baseline --ppo-nsteps 0 --spider-steps 0 --recurrent-steps 0"
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  echo "pass $i: $set" | tee -a $O/progress.log
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o q -- $B > $O/p$i.log 2>&1 || { echo "pass $i failed" | tee -a $O/progress.log; grep -m2 -i "exceeds\|error code" $O/p$i.log; continue; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/qpmc3/p*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        if "sumo_rollout_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k, v in sorted(acc.items()):
        ids = sorted(v)[1:]
        print(d.split("/")[-2], k, "per launch: %.4g | per env step: %.1f" % (sum(v[i] for i in ids) / len(ids), sum(v[i] for i in ids) / len(ids) / (4096 * 20)))
PY
rm -rf $O/p?
