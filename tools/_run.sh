python tools/ssfeat_check.py 2>&1 | grep "N=" | head -12
python -m pytest tests/test_gpu_families.py -m gpu -x -q 2>&1 | tail -2
LC_VARIANT_REPEAT=2 python tools/variants.py run --fam ng --iters 12 fused_v2
LC_VARIANT_REPEAT=2 python tools/variants.py run --fam eg --iters 12 fused_v2
python tools/fuzz_parity.py 300 211 2>&1 | tail -1
