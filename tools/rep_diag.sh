echo "nproc: $(nproc); cpu_count: $(python3 -c 'import os; print(os.cpu_count(), len(os.sched_getaffinity(0)))')"
for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /sys/fs/cgroup/cpu/cpu.stat; do echo "== $f"; cat $f 2>/dev/null; done
cat /proc/self/cgroup
for i in 1 2 3 4; do DIAG_POLL=0 python3 tools/bench_warm_diag.py --warmup 5 --steps 20 --quiet 2>/dev/null; grep -h thrott /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.stat 2>/dev/null | tr '\n' ' '; echo; done
echo "--- OMP_NUM_THREADS=1 etc"
for i in 1 2 3 4 5 6; do OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1 DIAG_POLL=0 python3 tools/bench_warm_diag.py --warmup 5 --steps 20 --quiet 2>/dev/null; grep -h thrott /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.stat 2>/dev/null | tr '\n' ' '; echo; done
python3 - <<'PY'
import os, torch, threading
print("threads after import torch:", len(os.listdir('/proc/self/task')), "torch threads", torch.get_num_threads())
torch.cuda.init(); x=torch.zeros(10,device='cuda'); torch.cuda.synchronize()
print("threads after cuda init:", len(os.listdir('/proc/self/task')))
PY
