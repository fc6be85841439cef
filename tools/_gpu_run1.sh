mkdir -p gpurun_out/r02
export DG_HOST_DEBUG=1
timeout 300 python tests/perf/host_path.py > gpurun_out/r02/host_path.txt 2>&1
DG_HOST_DIRECT=0 timeout 300 python tests/perf/host_path.py > gpurun_out/r02/host_path_staged.txt 2>&1
unset DG_HOST_DEBUG
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import dgtest as T
V,F=T.icosphere(71); T.write_obj('/tmp/ico71.obj',V,F)
PY
timeout 300 tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 5 > gpurun_out/r02/addfunction_256.json 2>&1
timeout 300 tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "128 128 128" 5 > gpurun_out/r02/addfunction_128.json 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_1.txt 2>&1
tail -5 gpurun_out/r02/pytest_gpu_1.txt
