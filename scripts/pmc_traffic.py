"""Turns the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_suite_r02.sh (gpurun_out/pmc_{FETCH,WRITE}_SIZE_*)
into profiles/<round>_pmc_traffic.md and the hero / cfg5 entries of profiles/traffic.json.

HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950
FETCH_SIZE reports exactly half of the bytes of 16-byte-per-lane reads (guides/MI355X_MICROARCH.md, HBM section;
re-confirmed in round 1 on sr_pack_nhwc_kernel: 8.6 MB streamed, FETCH_SIZE = 4205 KiB, WRITE_SIZE = 8400 KiB).
Infinity-Cache hits are included in FETCH_SIZE, so this is fabric traffic, an upper bound on DRAM traffic."""
import collections, csv, glob, json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def agg(tag, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(R, "gpurun_out", f"pmc_{counter}_{tag}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                d[r["Kernel_Name"]][0] += 1
                d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


def main(rnd):
    path = os.path.join(R, "profiles", "traffic.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    lines = []
    for wl, tag, kerns in (("hero_cfg3", "hero", {"hero_cfg3:sr_wino_kernel": "sr_wino_kernel<2, true, true", "hero_cfg3:mlp_sweep": "sr_mlp_volume_kernel",
                                              "hero_cfg3:sr_mlp_volume_kernel": "sr_mlp_volume_kernel", "hero_cfg3:sr_wino4_kernel": "sr_wino4_kernel",
                                              "hero_cfg3:sr_wino4ws_kernel": "sr_wino4ws_kernel"}),
                           ("hero_cfg5_volume", "cfg5", {"hero_cfg5_volume": "sr_mlp_volume_kernel"})):
        f, w = agg(tag, "FETCH_SIZE"), agg(tag, "WRITE_SIZE")
        if not f:
            continue
        combined = {}
        for key in kerns:   # (a kernel the step no longer launches must not keep a stale entry)
            out.pop(key, None)
        lines.append(f"## {wl}\n\n| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM-side bytes/launch (2F+W)*1024 |\n|---|---|---|---|---|")
        for rank, (k, (n, v)) in enumerate(sorted(f.items(), key=lambda kv: -kv[1][1])):
            wn, wv = w.get(k, [0, 0.0])
            b = (2 * v / n + wv / max(wn, 1)) * 1024
            if rank < 14 or any(sub in k for sub in kerns.values()):
                lines.append(f"| `{k[:70]}` | {n} | {v/n:.1f} | {wv/max(wn,1):.1f} | {b/1e6:.1f} MB |")
            for key, sub in kerns.items():
                if sub in k:   # several instantiations of one kernel (r06: sr_wino4ws_kernel<GENERIC_ACT, RES>): launch-weighted mean
                    m = re.search(r"sr_\w+", k)
                    acc = combined.setdefault(key, [0, 0.0, m.group(0) if m else k[:60], set()])
                    acc[0] += n
                    acc[1] += b * n
                    acc[3].add(k[:90])
        for key, (cn, cb, cname, inst) in combined.items():
            if cn:
                one = re.search(r"sr_\w+(<[^(]*>)?", sorted(inst)[0]).group(0) if len(inst) == 1 else cname
                out[key] = {"bytes": cb / cn, "kernel": one, "launches": cn}
        lines.append("")
    json.dump(out, open(path, "w"), indent=1)
    open(os.path.join(R, "profiles", f"{rnd}_pmc_traffic.md"), "w").write(
        "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), MI355X\n\n" + __doc__ + "\n\n" + "\n".join(lines) +
        "\n(dot-product sweep: see " + rnd + "_pmc_dot.md)\n")
    print("\n".join(lines)); print(out)


main(sys.argv[1] if len(sys.argv) > 1 else "r02")
