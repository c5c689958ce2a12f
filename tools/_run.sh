mkdir -p gpurun_out/r04h
python -m pytest tests/test_gpu_families.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r04h/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r04h/pytest.log
tail -3 gpurun_out/r04h/pytest.log
LC_VARIANT_REPEAT=3 python tools/variants.py run --fam ng --iters 12 fused_v2 > gpurun_out/r04h/var_ns.log 2>&1
cat gpurun_out/r04h/var_ns.log
LC_VARIANT_REPEAT=2 python tools/variants.py run --fam eg --iters 12 fused_v2
