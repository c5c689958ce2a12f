#!/bin/bash
# The rocprofv3 evidence of a round, every configuration bench.py reports (on the GPU box, from the repo root):
#   bash tools/profile_all.sh r05        -> gpurun_out/prof_r05_<tag>/...; summarise here with
#   for t in northstar config2 ...; do python tools/summarize_prof.py gpurun_out/prof_r05_$t > profiles/r05_${t}_rocprof_summary.txt; done
rnd=${1:-r05}
bash tools/profile_round.sh ${rnd}_northstar
bash tools/profile_round.sh ${rnd}_config2 --config 2 --steps 200 --warmup 20
bash tools/profile_round.sh ${rnd}_config5 --config 5
bash tools/profile_round.sh ${rnd}_dgmm --config dgmm --steps 20 --warmup 3
bash tools/profile_round.sh ${rnd}_bemm --config bemm --steps 20 --warmup 3
bash tools/profile_round.sh ${rnd}_wide256 --config wide256
bash tools/profile_round.sh ${rnd}_d96 --config d96
bash tools/profile_round.sh ${rnd}_k20 --config k20
bash tools/profile_round.sh ${rnd}_k40 --config k40
bash tools/profile_round.sh ${rnd}_d32 --config d32 --steps 20 --warmup 3
bash tools/profile_round.sh ${rnd}_d48 --config d48 --steps 20 --warmup 3
bash tools/profile_round.sh ${rnd}_k8 --config k8 --steps 20 --warmup 3
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/prof_${rnd}_learn; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 tools/learn_bench.py 10000000 64 32 > $out/kt.log 2>&1
out=gpurun_out/prof_${rnd}_small_d; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 tools/small_d_probe.py 2 4 > $out/kt.log 2>&1
