"""gpurun_out/pmc_<tag>N (scripts/pmc_dot.sh passes) -> profiles/<round>_pmc_dot.md + dot entries of profiles/traffic.json.

HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (both counters are KiB; FETCH_SIZE counts half of the
bytes of 16-byte-per-lane reads on gfx950, guides/MI355X_MICROARCH.md).  Counter values are per-launch averages."""
import collections, csv, glob, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
K, Cc, h, w, D = 7, 16, 120, 160, 64
ALG = lambda B: B * (4 * ((K + 1) * Cc * h * w + D * h * w + h * w) + 4 * (32 * K + 16 + D))
RUNS = [("ldsB8", "LDS-staged sweep, batch 8 (`dot_b8`)", 8), ("ldsB1", "LDS-staged sweep, batch 1 (`dot_cfg2`)", 1),
        ("l1qB8", "L1-gather sweep of round 1 (`SR_DOT_LDS=0`), batch 8", 8)]


def collect(tag):
    out, name = {}, None
    for d in sorted(glob.glob(os.path.join(R, "gpurun_out", f"pmc_{tag}?"))):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            agg, n = collections.defaultdict(float), collections.Counter()
            for r in csv.DictReader(open(f)):
                if "sr_dot_volume" in r["Kernel_Name"] and "lowest" not in r["Kernel_Name"]:
                    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            for k, v in agg.items():
                out[k] = v / n[k]
    return name, out


lines = ["# rocprofv3 --pmc passes of the dot-product sweep kernels (MI355X; per-launch averages; one pass per counter group)\n",
         __doc__, ""]
traffic_path = os.path.join(R, "profiles", "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
for tag, title, B in RUNS:
    name, c = collect(tag)
    if not c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0      # the counter sums the 8 XCDs
    waves = c["SQ_WAVES"]
    alg = ALG(B)
    hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    lines += [f"## {title}: `{name}`", "",
              "| quantity | value |", "|---|---|",
              f"| kernel duration (GRBM_GUI_ACTIVE / 8 XCDs) | {cyc:,.0f} cycles |",
              f"| waves | {waves:,.0f} |",
              f"| instructions per wave: VALU / SALU / LDS / VMEM read / SMEM | {c['SQ_INSTS_VALU']/waves:,.0f} / {c['SQ_INSTS_SALU']/waves:,.0f} / {c['SQ_INSTS_LDS']/waves:,.0f} / {c['SQ_INSTS_VMEM_RD']/waves:,.0f} / {c['SQ_INSTS_SMEM']/waves:,.0f} |",
              f"| wave time: issuing / parked (s_waitcnt, barrier) / issue-stalled | {100*c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']:.0f} % / {100*c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.0f} % / {100*c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.0f} % |",
              f"| LDS pipe active (SQ_LDS_IDX_ACTIVE / 256 CUs / duration) | {100*c['SQ_LDS_IDX_ACTIVE']/256/cyc:.0f} % |",
              f"| LDS bank-conflict cycles / LDS active cycles | {100*c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):.0f} % |",
              f"| TA busy (TA_BUSY_avr / duration) | {100*c['TA_BUSY_avr']/cyc:.0f} % |",
              f"| vector-L1 accesses (TCP_TOTAL_CACHE_ACCESSES) / L1->L2 read requests | {c['TCP_TOTAL_CACHE_ACCESSES_sum']:,.0f} / {c['TCP_TCC_READ_REQ_sum']:,.0f} |",
              f"| L2 hit rate | {100*c['TCC_HIT_sum']/(c['TCC_HIT_sum']+c['TCC_MISS_sum']):.0f} % |",
              f"| FETCH_SIZE / WRITE_SIZE (KiB) | {c['FETCH_SIZE']:,.0f} / {c['WRITE_SIZE']:,.0f} |",
              f"| HBM-side bytes per launch (2F+W)*1024 | {hbm/1e6:.1f} MB |",
              f"| algorithmic bytes per launch | {alg/1e6:.1f} MB |",
              f"| traffic / algorithmic | {hbm/alg:.2f} |", ""]
    if tag.startswith("lds"):
        traffic["dot_b8" if B == 8 else "dot_cfg2"] = {"bytes": hbm, "algorithmic_bytes": alg, "kernel": name}
json.dump(traffic, open(traffic_path, "w"), indent=1)
open(os.path.join(R, "profiles", f"{rnd}_pmc_dot.md"), "w").write("\n".join(lines))
print("\n".join(lines))
