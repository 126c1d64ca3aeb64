"""Decodes a -DSR_WINO_TRACE dump of sr_wino8_kernel: stamps 2k / 2k+1 = wave 0 at the top of slab k / in front of its barrier.
usage: python scripts/wino8_trace.py gpurun_out/wino8_trace.bin"""
import sys
import numpy as np

raw = open(sys.argv[1], "rb").read()
blocks, R, E, chunks = np.frombuffer(raw[:16], np.int32)
t = np.frombuffer(raw[16:], np.uint64).reshape(blocks, R * E).astype(np.int64)
n = (R * E) // 2
top, bar = t[:, 0:2 * n:2], t[:, 1:2 * n:2]
ok = (top > 0) & (bar > 0)
n = int(ok.sum(axis=1).min())           # slabs every block stamped
n -= n % chunks
print(f"blocks {blocks}, slabs/region {chunks}, slabs stamped by every block {n}")
top, bar = top[:, :n], bar[:, :n]
issue = bar - top                       # top of slab -> in front of the barrier
wait = top[:, 1:] - bar[:, :-1]         # barrier (+ the region bookkeeping between slabs)
period = top[:, 1:] - top[:, :-1]
k = np.arange(n)
print("slab-in-region  issue(mean p10 p90)        barrier+gap(mean p10 p90)   period")
for c in range(chunks):
    sel = (k % chunks == c) & (k >= chunks) & (k < n - 1)
    a = issue[:, sel].ravel(); b = wait[:, sel[:-1]].ravel(); pr = period[:, sel[:-1]].ravel()
    print(f"   {c:2d}          {a.mean():8.0f} {np.percentile(a,10):8.0f} {np.percentile(a,90):8.0f}      "
          f"{b.mean():8.0f} {np.percentile(b,10):8.0f} {np.percentile(b,90):8.0f}     {pr.mean():8.0f}")
reg = top[:, chunks::chunks]
rp = (reg[:, 1:] - reg[:, :-1]).ravel()
print(f"region period mean {rp.mean():.0f} (ideal 32 MFMA x 64 clk x 2 waves x {chunks} slabs = {4096 * chunks})")
