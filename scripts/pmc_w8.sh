#!/bin/bash
# Wave-time / wait / instruction counters of sr_wino8_kernel (full and ablated builds) on the 64 -> 64 @ 8x240x320 layer.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
export SR_MICRO_SHAPES=0 SR_MICRO_MODES=2 SR_WINO8=2
for lib in full abl2 abl8 abl392; do
  if [ $lib != full ]; then export SR_HIP_LIBRARY=$R/simplerecon_amd/abl/lib_$lib.so; else unset SR_HIP_LIBRARY; fi
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
             "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" \
             "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_w8_${lib}_$i -o w -- python $R/scripts/wino8_micro.py > $O/pmc_w8_${lib}_$i.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
for lib in ("full", "abl2", "abl8", "abl392"):
    agg = collections.defaultdict(float); n = 0
    for d in sorted(glob.glob("$O/pmc_w8_%s_?" % lib)):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            disp = set()
            for r in csv.DictReader(open(f)):
                if "wino8" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
            n = max(n, len(disp))
    print(lib, "launches", n)
    wc = agg.get("SQ_WAVE_CYCLES", 1.0) / 3.0   # counted in all three passes
    for k in sorted(agg):
        v = agg[k] / (3.0 if k == "SQ_WAVE_CYCLES" else 1.0)
        print(f"   {k:32s} {v / max(n,1):16.0f} per launch   {v / wc:8.4f} of wave cycles")
PY
