mkdir -p gpurun_out/r04e
python tools/ssfeat_check.py > gpurun_out/r04e/ssfeat.log 2>&1
tail -45 gpurun_out/r04e/ssfeat.log
python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/r04e/bench5.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r04e/bench5.log') if x.startswith('{')][-1]
d=json.loads(l); print(d['ms_per_step'], d['kernels'], d['roofline']['estep_frac'], d['roofline']['suffstat_frac'])
PY
