cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/att
rocprofv3 --help 2>&1 | grep -i -A2 "att\|pc-sampling\|pc_sampling" | head -40 > gpurun_out/att/help.txt
timeout 300 rocprofv3 --att --att-target-cu 1 --kernel-include-regex "estep_kernel" -d gpurun_out/att/out -- python3 bench.py --rows 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-other-configs > gpurun_out/att/att.log 2>&1
echo "att rc=$?" >> gpurun_out/att/att.log
tail -5 gpurun_out/att/att.log
ls -R gpurun_out/att/out 2>/dev/null | head -20
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100 -d gpurun_out/att/pcs -- python3 bench.py --rows 2000000 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-other-configs > gpurun_out/att/pcs.log 2>&1
echo "pcs rc=$?" >> gpurun_out/att/pcs.log
tail -5 gpurun_out/att/pcs.log
ls -R gpurun_out/att/pcs 2>/dev/null | head
