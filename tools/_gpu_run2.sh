mkdir -p gpurun_out/r02
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_2.txt 2>&1
tail -8 gpurun_out/r02/pytest_gpu_2.txt
timeout 600 python bench.py > gpurun_out/r02/bench1.json 2> gpurun_out/r02/bench1.err
tail -c 3000 gpurun_out/r02/bench1.json; tail -5 gpurun_out/r02/bench1.err
