"""Per-queue view of one step from a rocprofv3 --kernel-trace CSV: for the queue that runs the MLP sweep (the main stream), its idle
intervals inside a step and which queues were busy meanwhile; per queue, busy time and first / last kernel relative to the sweep.
    python scripts/timeline_streams.py <kernel_trace.csv> [step_index]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
qk = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44], r.get(qk, "?")) for r in rows))
sweeps = [i for i, k in enumerate(ks) if "sr_mlp_volume_kernel" in k[2]]
step = int(sys.argv[2]) if len(sys.argv) > 2 else len(sweeps) - 3
# a step = from the first kernel after the previous step's last kernel ... use sweep-to-sweep windows shifted to the stem of the main queue
main_q = ks[sweeps[step]][3]
t_sweep = ks[sweeps[step]][0]
t_next = ks[sweeps[step + 1]][0]
period = t_next - t_sweep
# window: [t_sweep - 4.5 ms, t_sweep - 4.5 ms + period)
stems = [k for k in ks if k[2].startswith("sr_stem_kernel") and k[0] < t_sweep and k[0] > t_sweep - period]
t0 = min(k[0] for k in stems) - 200000
t1 = t0 + period
sel = [k for k in ks if t0 <= k[0] < t1]
print(f"step window {period / 1e6:.3f} ms; main queue {main_q}; sweep starts at +{(t_sweep - t0) / 1e6:.3f} ms, ends +{(ks[sweeps[step]][1] - t0) / 1e6:.3f} ms")
byq = collections.defaultdict(list)
for k in sel:
    byq[k[3]].append(k)
for q, lst in sorted(byq.items(), key=lambda kv: -sum(k[1] - k[0] for k in kv[1])):
    busy = sum(k[1] - k[0] for k in lst)
    print(f"  queue {q}: {len(lst):4d} kernels, {busy / 1e6:7.3f} ms of kernel time, first +{(lst[0][0] - t0) / 1e6:.3f} ms ({lst[0][2]}), last ends +{(max(k[1] for k in lst) - t0) / 1e6:.3f} ms ({lst[-1][2]})")
main = sorted(byq[main_q])
print("main-queue idle intervals > 15 us (start, length, kernel before -> after, kernels of other queues running in it):")
end = main[0][1]; prev = main[0][2]
for s, e, name, q in main[1:]:
    if s - end > 15000:
        others = [k for k in sel if k[3] != main_q and k[1] > end and k[0] < s]
        oq = collections.Counter(k[3] for k in others)
        print(f"  +{(end - t0) / 1e6:7.3f} ms  {(s - end) / 1e3:8.1f} us   {prev} -> {name}   others: {dict(oq)}")
    if e > end:
        end, prev = e, name
