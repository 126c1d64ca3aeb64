#!/bin/bash
# Instruction mix / waits of sr_mlp_volume_kernel (hero sweep, batch 8): per-MFMA VALU / LDS / VMEM counts for the pipe model of DESIGN 3.3c.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_mlp_$i -o m -- python $R/scripts/mlp_micro.py > $O/pmc_mlp_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for d in sorted(glob.glob("$O/pmc_mlp_?")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:44]
            if "mlp_volume" in k:
                key = (k, r["Grid_Size"])
                agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key].add((d, r["Dispatch_Id"]))
for key, v in agg.items():
    launches = len(n[key]) / 3.0
    wc = v["SQ_WAVE_CYCLES"] / 3.0
    mfma = v["SQ_INSTS_VALU_MFMA_MOPS_F32"] / 4.0   # 32x32x2 f32 = 2048 MACs = 4 "MOPS" units of 512
    print(key, "launches", launches)
    for c in sorted(v):
        x = v[c] / (3.0 if c == "SQ_WAVE_CYCLES" else 1.0)
        print(f"   {c:32s} {x / launches:16.0f} per launch  {x / wc:8.4f} of wave cycles  {x / max(mfma,1):8.3f} per MFMA")
PY
