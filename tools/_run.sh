mkdir -p gpurun_out/r04y
python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/r04y/bench5.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r04y/bench5.log') if x.startswith('{')][-1]
d=json.loads(l); print(d['ms_per_step'], d['kernels']['estep_ms'], d['kernels']['suffstat_ms'], d['roofline']['estep_frac'], d['roofline']['suffstat_frac'])
PY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
python tools/ssfeat_check.py 2>&1 | tail -20
(python tools/fuzz_parity.py 1200 101; LC_FUZZ_CACHE=1 python tools/fuzz_learn.py 500 103; python tools/fuzz_learn.py 500 107; python tools/fuzz_kernels.py 300 109) > gpurun_out/r04y/fuzz.log 2>&1
tail -12 gpurun_out/r04y/fuzz.log
