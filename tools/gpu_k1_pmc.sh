#!/bin/bash
# K1 variant measurement on the GPU box: bench of both variants + PMC passes (counters only) of the filtered one.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-ab}
for fast in 0 1; do
  DG_FORCE="k1_fast=$fast" timeout 300 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/${TAG}_bench_fast$fast.json 2> gpurun_out/${TAG}_bench_fast$fast.err
  python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_fast$fast.json"))
print("k1_fast=$fast", d["value"], "Mnodes/s", d["ms_per_step"], "ms")
PY
done
export DG_FORCE="k1_fast=1"
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extras > /tmp/prof_kt.log 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d /tmp/prof_$TAG -o pmc$i -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-extras > /tmp/prof_pmc$i.log 2>&1
done
cd "$GRAFT_REPO_ROOT"
python - <<PY > gpurun_out/${TAG}_pmc.txt
import sqlite3, glob
db = glob.glob("/tmp/prof_$TAG/**/kt_results.db", recursive=True)[0]
c = sqlite3.connect(db)
print("# kernel durations (rocprofv3 --kernel-trace --stats), k1_fast=1")
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if "dg::" in name: print("%-70s calls %3d avg %.3f ms" % (name[:70], calls, avg / 1e3))
for r in c.execute("select name,duration,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size from kernels where name like '%k_sample_fast%' limit 1"):
    print("#", r)
print("# PMC, mean per dispatch")
for i in (1, 2, 3):
    for db in glob.glob("/tmp/prof_$TAG/**/pmc%d_results.db" % i, recursive=True):
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        q = None
        if "counters_collection" in tabs:
            q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
        if q:
            for r in c.execute(q):
                if "k_sample" in str(r[0]) or "k_heavy" in str(r[0]):
                    nm = "k_sample_fast" if "k_sample_fast" in str(r[0]) else ("k_heavy_subtrees" if "subtrees" in str(r[0]) else ("k_heavy_finish" if "finish" in str(r[0]) else "k_sample_nodes"))
                    print("%-20s %-28s %16.6g %4d" % (nm, r[1], r[2], r[3]))
        else:
            print("# tables:", tabs)
PY
cat gpurun_out/${TAG}_pmc.txt
