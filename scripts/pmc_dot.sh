#!/bin/bash
# SQ / LDS / TA counters of the dot-product sweep kernels (one --pmc pass per counter group; no tracing flags).
# usage: scripts/pmc_dot.sh <tag>   (kernel variant comes from the SR_DOT_* environment)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-dot}; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_${TAG}$i -o w -- python $R/scripts/dot_micro.py > $O/pmc_${TAG}$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/pmc_${TAG}?")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            if "dot_volume" in k:
                print("$TAG", d[-1], k, {a: f"{b / n[(k, a)]:.4g}" for a, b in v.items()})
PY
