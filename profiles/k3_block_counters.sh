#!/bin/bash
# K3 block shapes by COUNTER, not by time: HBM fetch bytes and L2 hit rate of k_density_cells at 256^3 per block shape
# (waves along x / y / z of the blocks consecutive wave ids fill).  One rocprofv3 --pmc pass per counter group and shape.
#   bash profiles/k3_block_counters.sh > profiles/r04_k3_blocks_pmc.txt        (on the GPU box, through gpurun)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=/tmp/k3blocks; mkdir -p $OUT
python -c "import torch" 2>/dev/null
echo "# k_density_cells, icosphere SDF 256^3, h = 0.1: per launch (mean of 2)"
echo "# block(x y z)   kernel_ms   HBM_fetch_GB   HBM_write_GB   L2_hit"
for shape in "1 16 8" "1 12 8" "1 8 8" "1 6 22" "1 3 43" "1 2 64" "1 8 16" "2 8 8"; do
  set -- $shape
  export DG_FORCE="k3_rb0=$1;k3_rb1=$2;k3_rb2=$3"
  tag=b$1_$2_$3
  timeout 120 rocprofv3 --kernel-trace --stats -d $OUT -o ${tag}_kt -- python profiles/pmc_workloads.py k3 > $OUT/${tag}_kt.log 2>&1
  # (FETCH_SIZE and WRITE_SIZE in ONE pass never finished on the box of round 4 -- every run sat out its timeout and returned
  # nothing; one counter per pass, like profiles/collect.sh)
  timeout 120 rocprofv3 --pmc FETCH_SIZE -d $OUT -o ${tag}_p1 -- python profiles/pmc_workloads.py k3 > $OUT/${tag}_p1.log 2>&1
  timeout 120 rocprofv3 --pmc WRITE_SIZE -d $OUT -o ${tag}_p3 -- python profiles/pmc_workloads.py k3 > $OUT/${tag}_p3.log 2>&1
  timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT -o ${tag}_p2 -- python profiles/pmc_workloads.py k3 > $OUT/${tag}_p2.log 2>&1
  python - "$OUT" "$tag" "$shape" <<'PY'
import glob, os, sqlite3, sys
d, tag, shape = sys.argv[1], sys.argv[2], sys.argv[3]
def db(stem):
    hits = sorted(glob.glob(os.path.join(d, "**", stem + "*.db"), recursive=True))
    return hits[-1] if hits else None
def counters(p):
    out = {}
    if not p: return out
    c = sqlite3.connect(p)
    for k, cn, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        if "k_density_cells" in str(k): out[cn] = v
    return out
ms = float("nan")
p = db(tag + "_kt")
if p:
    c = sqlite3.connect(p)
    for name, avg in c.execute("select name, average from top_kernels"):
        if "k_density_cells" in str(name): ms = avg * 1e-3 if avg > 1e4 else avg
c1, c2 = counters(db(tag + "_p1")), counters(db(tag + "_p2"))
c1.update(counters(db(tag + "_p3")))
f = 2.0 * c1.get("FETCH_SIZE", float("nan")) * 1024 / 1e9
w = c1.get("WRITE_SIZE", float("nan")) * 1024 / 1e9
h, m = c2.get("TCC_HIT_sum", float("nan")), c2.get("TCC_MISS_sum", float("nan"))
print("%-14s %10.1f %14.1f %14.2f %8.3f" % (shape, ms, f, w, h / (h + m) if h == h and (h + m) > 0 else float("nan")))
PY
done
