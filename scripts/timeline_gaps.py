"""GPU idle time inside the timed steps from a rocprofv3 --kernel-trace CSV: union of the kernel intervals of all streams, the gaps
between them, and what ran before / after the largest ones.
    python scripts/timeline_gaps.py <kernel_trace.csv> [steps_to_skip]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows), key=lambda t: t[0])
# steps: delimited by the MLP sweep launches (one per step)
sweeps = [i for i, k in enumerate(ks) if k[2].startswith("void sr_mlp_volume_kernel") or k[2].startswith("sr_mlp_volume_kernel")]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if len(sweeps) < skip + 2:
    raise SystemExit(f"only {len(sweeps)} sweeps in the trace")
t0, t1 = ks[sweeps[skip]][0], ks[sweeps[-1]][0]
n_steps = len(sweeps) - 1 - skip
sel = [k for k in ks if k[0] >= t0 and k[0] < t1]
busy, gaps, cur_end, last = 0, [], None, None
for s, e, name in sel:
    if cur_end is None:
        cur_s, cur_end, last = s, e, name
        continue
    if s > cur_end:
        busy += cur_end - cur_s
        gaps.append((s - cur_end, last, name))
        cur_s, cur_end = s, e
        last = name
    else:
        if e > cur_end:
            cur_end, last = e, name
busy += cur_end - cur_s
span = t1 - t0
print(f"{n_steps} steps, {span / n_steps / 1e6:.3f} ms per step; some kernel running {busy / span * 100:.2f} % of the time; "
      f"{len(gaps) / n_steps:.0f} gaps per step, {sum(g[0] for g in gaps) / n_steps / 1e3:.1f} us idle per step")
hist = {}
for g, a, b in gaps:
    key = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<30us" if g < 30000 else ">=30us"
    h = hist.setdefault(key, [0, 0]); h[0] += 1; h[1] += g
for k_ in ("<2us", "<5us", "<10us", "<30us", ">=30us"):
    if k_ in hist:
        print(f"  gaps {k_:6s}: {hist[k_][0] / n_steps:7.1f} per step, {hist[k_][1] / n_steps / 1e3:8.1f} us per step")
print("largest gaps (us, kernel before -> kernel after):")
for g, a, b in sorted(gaps, key=lambda t: -t[0])[:25]:
    print(f"  {g / 1e3:8.1f}  {a}  ->  {b}")
