mkdir -p gpurun_out/r04z
python bench.py --steps 20 --warmup 5 > gpurun_out/r04z/bench_line.json 2> gpurun_out/r04z/bench_err.log
tail -c 200 gpurun_out/r04z/bench_line.json
python -m pytest tests -m gpu -x -q > gpurun_out/r04z/pytest_full.log 2>&1; echo "rc=$?" >> gpurun_out/r04z/pytest_full.log
tail -3 gpurun_out/r04z/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
