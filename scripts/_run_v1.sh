set -e
mkdir -p gpurun_out/r05q
python -c "import __graft_entry__ as e; e.smoke()" > gpurun_out/r05q/smoke.txt 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/r05q/gpu_tests.txt 2>&1 || { tail -40 gpurun_out/r05q/gpu_tests.txt; exit 1; }
tail -5 gpurun_out/r05q/gpu_tests.txt
python bench.py > gpurun_out/r05q/bench_default.json 2> gpurun_out/r05q/bench_default.err
tail -1 gpurun_out/r05q/bench_default.json
python scripts/bench_sparse.py --no-cpu-baseline > gpurun_out/r05q/sparse.json 2> gpurun_out/r05q/sparse.err
tail -1 gpurun_out/r05q/sparse.json
