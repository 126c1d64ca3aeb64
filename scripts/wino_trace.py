"""Decodes a -DSR_WINO_TRACE dump (SR_WINO_TRACE_FILE): per-workgroup phase timeline of sr_wino_kernel.
usage: python scripts/wino_trace.py gpurun_out/wino_trace.bin"""
import sys
import numpy as np

raw = open(sys.argv[1], "rb").read()
blocks, R, E, chunks = np.frombuffer(raw[:16], np.int32)
t = np.frombuffer(raw[16:], np.uint64).reshape(blocks, R, E).astype(np.int64)
hw = t[:, 0, 15]
xcc, hwid = hw >> 32, hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF
se = (hwid >> 13) & 0x7  # gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
sh = (hwid >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 50 + cu
t0 = t[:, 0, 0].min()
print(f"blocks {blocks}, chunks/region {chunks}; distinct (xcc,se,sh,cu) keys: {len(set(key.tolist()))}")
names = ["start", "slab0"] + [f"{p}{c}" for c in range(5) for p in ("T", "M")] + ["loopend", "Oex", "end"]
# phase durations averaged over blocks / regions 1..R-2
valid = t[:, 1:R - 1, :]
def dur(a, b):
    d = (valid[:, :, b] - valid[:, :, a]).ravel()
    d = d[(valid[:, :, a].ravel() > 0) & (valid[:, :, b].ravel() > 0)]
    return d.mean(), np.percentile(d, 10), np.percentile(d, 90)
print("phase durations in shader clocks (mean, p10, p90):")
print("  start->slab0 stored   ", dur(0, 1))
for c in range(min(chunks, 5)):
    prev = 1 if c == 0 else 3 + 2 * (c - 1)
    print(f"  chunk {c}: T (prev M end -> T barrier)", dur(prev, 2 + 2 * c), " M issue", dur(2 + 2 * c, 3 + 2 * c))
last = 3 + 2 * (min(chunks, 5) - 1)
print("  last M -> loop end    ", dur(last, 12))
print("  loop end -> O exchanged", dur(12, 13))
print("  O exchanged -> stores issued + barrier", dur(13, 14))
print("  epilogue detail: loopend->rv loads issued", dur(12, 10), " ->O writes issued", dur(10, 11), " ->barrier", dur(11, 13),
      " ->transform+stores issued", dur(13, 15), " ->final barrier", dur(15, 14))
print("  region total           ", dur(0, 14))
per = (valid[:, 1:, 0] - valid[:, :-1, 0]).ravel()
print("  region period          ", per[per > 0].mean())
# co-resident pairs: print the timelines of two workgroups on one CU
from collections import defaultdict
groups = defaultdict(list)
for bi in range(blocks):
    groups[int(key[bi])].append(bi)
shown = 0
for k, bl in sorted(groups.items()):
    if len(bl) >= 2 and shown < 3:
        shown += 1
        print(f"CU key {k}: workgroups {bl}")
        for bi in bl[:2]:
            for r in range(2, 5):
                ev = t[bi, r]
                print(f"   wg {bi} region {r}: " + " ".join(f"{names[e]}={int(ev[e] - t0)}" for e in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 14) if ev[e] > 0))
print("workgroups per CU key histogram:", np.bincount([len(v) for v in groups.values()]))
