mkdir -p gpurun_out/r04g
python -m pytest tests/test_gpu_families.py -m gpu -x -q > gpurun_out/r04g/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r04g/pytest.log
tail -3 gpurun_out/r04g/pytest.log
LC_VARIANT_REPEAT=2 python tools/variants.py run --fam ng --iters 12 fused_v2 > gpurun_out/r04g/var_ng.log 2>&1
LC_VARIANT_REPEAT=2 python tools/variants.py run --fam eg --iters 12 fused_v2 > gpurun_out/r04g/var_eg.log 2>&1
cat gpurun_out/r04g/var_ng.log gpurun_out/r04g/var_eg.log
