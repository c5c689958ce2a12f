for i in 1 2 3; do python tools/_ms.py 2>&1 | grep MS; done
python -m pytest tests/test_gpu_splitsearch.py -m gpu -x -q 2>&1 | tail -2
