"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gpurun_out/pmc_*) into profiles/traffic.json.

HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950
FETCH_SIZE reports exactly half of the bytes of 16-byte-per-lane reads (guides/MI355X_MICROARCH.md §HBM;
re-confirmed here on sr_pack_nhwc_kernel: 8.6 MB streamed, FETCH_SIZE = 4205 KiB, WRITE_SIZE = 8400 KiB).
Infinity-Cache hits are included in FETCH_SIZE, so this is fabric traffic, an upper bound on DRAM traffic."""
import collections, csv, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]][0] += 1
            d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d

def main(tag_round):
    out, lines = {}, []
    for wl, kern, fp, wp in (
            ("hero_cfg3", "sr_wino_kernel<2, true, true>", "pmc_fetch/hero_counter_collection.csv", "pmc_write/hero_counter_collection.csv"),
            ("dot_cfg2", "sr_dot_volume_kernel16q", "pmc_fetch_dot/dot_counter_collection.csv", "pmc_write_dot/dot_counter_collection.csv")):
        f = agg(os.path.join(R, "gpurun_out", fp), "FETCH_SIZE")
        w = agg(os.path.join(R, "gpurun_out", wp), "WRITE_SIZE")
        lines.append(f"## {wl}\n\n| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM-side bytes/launch (2F+W)*1024 |\n|---|---|---|---|---|")
        for k, (n, v) in sorted(f.items(), key=lambda kv: -kv[1][1])[:8]:
            wn, wv = w.get(k, [0, 0.0])
            b = (2 * v / n + wv / max(wn, 1)) * 1024
            lines.append(f"| `{k[:70]}` | {n} | {v/n:.1f} | {wv/max(wn,1):.1f} | {b/1e6:.1f} MB |")
            if kern in k:
                out[wl] = b
        lines.append("")
    json.dump(out, open(os.path.join(R, "profiles", "traffic.json"), "w"), indent=1)
    open(os.path.join(R, "profiles", f"{tag_round}_pmc_traffic.md"), "w").write(
        "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), MI355X\n\n" + __doc__ + "\n\n" + "\n".join(lines))
    print(out)

main(sys.argv[1] if len(sys.argv) > 1 else "r01")
