python -m pytest tests/test_gpu_parity.py tests/test_gpu_families.py tests/test_gpu_splitsearch.py -m gpu -x -q 2>&1 | tail -2
LC_VARIANT_REPEAT=3 python tools/variants.py run --iters 12 pre_fmax
LC_VARIANT_REPEAT=2 python tools/variants.py run --fam ng --iters 12 pre_fmax
LC_VARIANT_REPEAT=2 python tools/variants.py run --shape "1000000,16,8" --iters 300 pre_fmax
