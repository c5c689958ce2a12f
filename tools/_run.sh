cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/profile_round.sh r04_ns > /dev/null 2>&1
bash tools/profile_round.sh r04_c2 --config 2 --steps 200 --warmup 20 > /dev/null 2>&1
bash tools/profile_round.sh r04_c5 --config 5 > /dev/null 2>&1
bash tools/profile_round.sh r04_dgmm --config dgmm > /dev/null 2>&1
bash tools/profile_round.sh r04_bemm --config bemm > /dev/null 2>&1
bash tools/profile_round.sh r04_wide --config wide256 > /dev/null 2>&1
mkdir -p gpurun_out/prof_r04_learn
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_learn/kt -o kt -- python3 tools/learn_bench.py 10000000 64 32 > gpurun_out/prof_r04_learn/kt.log 2>&1
mkdir -p gpurun_out/r04z
python bench.py --steps 20 --warmup 5 > gpurun_out/r04z/bench_line.json 2> gpurun_out/r04z/bench_err.log
tail -c 300 gpurun_out/r04z/bench_line.json
python -m pytest tests -m gpu -x -q > gpurun_out/r04z/pytest_full.log 2>&1; echo "rc=$?" >> gpurun_out/r04z/pytest_full.log
tail -3 gpurun_out/r04z/pytest_full.log
