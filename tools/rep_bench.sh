# development: is the short-window bench reproducible?  (driver flags, fresh process each time)
rm -f gpurun_out/r2_rep.txt
for i in 1 2 3 4 5 6 7 8; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --ppo-nsteps 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('run', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['host'])
" >> gpurun_out/r2_rep.txt; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_rep_full1.json 2>/dev/null
python3 bench.py --gpus 1 --steps 100 --warmup 100 > gpurun_out/r2_rep_full2.json 2>/dev/null
BENCH_BACKEND=gloo python3 bench.py --gpus 2 --envs 1024 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_rep_2rank.json 2>gpurun_out/r2_rep_2rank.err
