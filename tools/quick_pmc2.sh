#!/bin/bash
# development: cache behaviour of the fused rollout launch (L1 / L2 hit rates)
export TMPDIR=/tmp
O=gpurun_out/qpmc2; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 20 --state-warmup 40 --no-cpu-baseline --ppo-nsteps 0 --spider-steps 0"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/a -o q -- $B > $O/a.log 2>&1 || exit 11
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $O/b -o q -- $B > $O/b.log 2>&1 || exit 12
python3 - <<'PY'
import csv, glob, collections
for d in ("a", "b"):
    f = glob.glob("gpurun_out/qpmc2/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "sumo_rollout_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k, v in acc.items():
        ids = sorted(v)[1:]
        print(k, "per env step: %.1f" % (sum(v[i] for i in ids) / len(ids) / (4096 * 20)))
PY
rm -rf $O/a $O/b
