for i in 1 2; do
python tools/_ms.py 2>&1 | grep MS
done
python -m pytest tests/test_gpu_splitsearch.py tests/test_a_multirank_gpu.py -m gpu -x -q 2>&1 | tail -2
LC_FUZZ_CACHE=1 timeout 600 python tools/fuzz_learn.py 100 29 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r04k
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04k/h4 -o h -- python3 tools/_ms.py > gpurun_out/r04k/h4.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r04k/h4/h_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)/1e6
print("total kernel ms (2 runs):", tot)
for r in rows[:8]:
    print("%-70s %6s %10.3f %9.4f %6s" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, r["Percentage"]))
PY
