#!/bin/bash
# rocprofv3 evidence for one bench configuration: kernel trace + stats, then three separate PMC passes
# (counters never share a run with the trace domains).  Usage (on the GPU box, from the repo root):
#   bash tools/profile_round.sh <tag> [bench.py args...]      -> gpurun_out/prof_<tag>/{kt,p1,p2,p3}
# Summarise with: python tools/summarize_prof.py gpurun_out/prof_<tag> > profiles/<round>_<tag>_rocprof_summary.txt
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
args="--steps 5 --warmup 1 --no-cpu-baseline --no-parity --no-other-configs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py $args > $out/kt.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/p1 -o p1 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py $args > $out/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/p2 -o p2 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- python3 bench.py $args > $out/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/p3 -o p3 --pmc WRITE_SIZE -- python3 bench.py $args > $out/p3.log 2>&1
find $out -name "*.csv" | head -20
