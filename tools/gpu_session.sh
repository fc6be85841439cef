#!/bin/bash
# one measuring session on the GPU box (run through gpurun): the legs named on the command line, outputs under gpurun_out/$TAG
#   bash tools/gpu_session.sh TAG leg [leg ...]     legs: addfn k3blocks k3pmc gputests bench multirank collect
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import torch" 2>/dev/null
for leg in "$@"; do
  case $leg in
    addfn)   # C++ addFunction(MeshSDF) at 256^3: return / device-ready / host-ready per chunk profile
      python - > /tmp/ico71.log 2>&1 <<'PY'
import sys; sys.path.insert(0, "tests"); import dgtest as T
V, F = T.icosphere(71); T.write_obj("/tmp/ico71.obj", V, F)
PY
      for prof in ${ADDFN_PROFILES:-"" "0.1,0.1,0.1,0.1,0.1,0.1,0.1,0.1,0.1,0.1" "0.14,0.13,0.12,0.11,0.10,0.10,0.09,0.08,0.07,0.06" "0.12,0.12,0.12,0.12,0.11,0.11,0.10,0.09,0.07,0.04" "0.18,0.17,0.16,0.14,0.12,0.10,0.07,0.04,0.02" "0.10,0.12,0.13,0.13,0.12,0.11,0.10,0.08,0.06,0.03,0.02"}; do
        unset DG_FORCE DG_FIELD_ONE_LAUNCH
        if [ "$prof" = one ]; then export DG_FIELD_ONE_LAUNCH=1; elif [ "$prof" = streams1 ]; then export DG_FORCE="field_streams=1"; elif [ -n "$prof" ]; then export DG_FORCE="field_fractions=$prof"; fi
        echo "profile '${prof:-default}':" >> $OUT/addfn.txt
        timeout 120 tests/cpp/build/unchanged_caller addfunction /tmp/ico71.obj "256 256 256" 5 >> $OUT/addfn.txt 2>> $OUT/addfn.err
        echo >> $OUT/addfn.txt
      done
      unset DG_FORCE DG_FIELD_ONE_LAUNCH
      cat $OUT/addfn.txt ;;
    k2band)  # parity of the band copy, then the config-5 digests through it
      timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "interpolate_vs_golden" > $OUT/k2band_parity.log 2>&1; tail -3 $OUT/k2band_parity.log
      timeout 900 python -m pytest tests/test_gpu_digests.py -x -q -m gpu -s -k "band_copy" > $OUT/k2band_digests.log 2>&1; grep "band_copy\|passed\|failed" $OUT/k2band_digests.log | tail -6 ;;
    k3blocks)
      timeout 400 python tools/k3_run.py --res 256 --steps 2 --check --sweep "k3_rb1=3,k3_rb2=43;k3_rb1=3,k3_rb2=129;k3_rb1=1,k3_rb2=129;k3_rb1=2,k3_rb2=129;k3_rb1=4,k3_rb2=32;k3_rb1=6,k3_rb2=22;k3_rb1=3,k3_rb2=65;k3_rb1=3,k3_rb2=26;k3_rb0=2,k3_rb1=3,k3_rb2=22" > $OUT/k3_blocks_256.jsonl 2> $OUT/k3_blocks.err
      cat $OUT/k3_blocks_256.jsonl ;;
    k3big)   # 512^3: parity of the point-lane kernel at offsets beyond 2^31, then its time
      timeout 900 python -m pytest tests/test_gpu_density_map.py -x -q -m gpu -k "beyond_two_gigabytes" > $OUT/k3big.log 2>&1; tail -4 $OUT/k3big.log
      timeout 600 python tools/k3_run.py --res 512 --steps 1 --sweep "k3_cells=1;k3_cells=0" > $OUT/k3_512.jsonl 2> $OUT/k3_512.err; cat $OUT/k3_512.jsonl ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
      timeout 600 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "cpp_multi_gpu_tool" > $OUT/tool.log 2>&1; tail -2 $OUT/tool.log ;;
    gputests)
      timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/gputests.log 2>&1; tail -5 $OUT/gputests.log ;;
    gputests_all)   # every GPU test, no stop at the first failure
      timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $OUT/gputests.log 2>&1; tail -30 $OUT/gputests.log ;;
    multirank)
      timeout 1800 python -m pytest tests/test_gpu_multirank.py -q -m gpu --timeout 600 > $OUT/multirank.log 2>&1; tail -40 $OUT/multirank.log ;;
    bench)
      timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json ;;
    collect)
      timeout 2400 bash profiles/collect.sh ${TAG} all > $OUT/collect.log 2>&1; tail -25 $OUT/collect.log ;;
    *) echo "unknown leg $leg" ;;
  esac
done
