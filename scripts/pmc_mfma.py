"""Turns the rocprofv3 --pmc MFMA pass of scripts/gpu_suite.sh (gpurun_out/pmc_mfma) into profiles/<round>_pmc_mfma_util.md.

SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 = executed fp32 MFMA FLOPs; SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe busy cycles summed
over the 1024 SIMDs (64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32); GRBM_GUI_ACTIVE is reported summed
over the 8 XCDs, so a kernel lasts GUI_ACTIVE / 8 cycles and MFMA utilisation = MFMA_BUSY / (1024 * GUI_ACTIVE / 8)."""
import collections, csv, glob, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
f = glob.glob(os.path.join(R, "gpurun_out", "pmc_mfma", "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[r["Kernel_Name"]] += 1
rows = []
for k, c in agg.items():
    if c["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0 or "sr_" not in k[:40]:
        continue
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    rows.append((c["GRBM_GUI_ACTIVE"], k, n[k], c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / n[k] / 1e9,
                 c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles)))
lines = ["# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES (one pass), hero_cfg3, MI355X",
         "", __doc__, "", "Command: `rocprofv3 --pmc ... -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline`", "",
         "| kernel | launches | executed MFMA GFLOP / launch | MFMA pipe utilisation |", "|---|---|---|---|"]
for _, k, cnt, gf, util in sorted(rows, reverse=True):
    lines.append(f"| `{k[:80]}` | {cnt} | {gf:.2f} | {100 * util:.1f} % |")
open(os.path.join(R, "profiles", f"{tag}_pmc_mfma_util.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-len(rows):]))
