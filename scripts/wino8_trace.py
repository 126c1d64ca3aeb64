"""Decodes a -DSR_WINO_TRACE dump of sr_wino8_kernel: four stamps per slab (wave 0): top of the slab, in front of its
barrier (after step 5), behind the barrier, after step 7.
usage: python scripts/wino8_trace.py gpurun_out/wino8_trace.bin"""
import sys
import numpy as np

raw = open(sys.argv[1], "rb").read()
blocks, R, E, chunks = np.frombuffer(raw[:16], np.int32)
t = np.frombuffer(raw[16:], np.uint64).reshape(blocks, R * E).astype(np.int64)
n = (R * E) // 4
st = t[:, :4 * n].reshape(blocks, n, 4)
ok = (st > 0).all(axis=2)
n = int(ok.sum(axis=1).min())
n -= n % chunks
st = st[:, :n]
print(f"blocks {blocks}, slabs/region {chunks}, slabs stamped by every block {n}")
s05 = st[:, :, 1] - st[:, :, 0]
bar = st[:, :, 2] - st[:, :, 1]
s67 = st[:, :, 3] - st[:, :, 2]
ctl = st[:, 1:, 0] - st[:, :-1, 3]
per = st[:, 1:, 0] - st[:, :-1, 0]
k = np.arange(n)
print("slab   steps 0-5   barrier   steps 6-7   control   period   (mean shader clocks; ideal 3072 / 0 / 1024 / 0 / 4096)")
for c in range(chunks):
    sel = (k % chunks == c) & (k >= chunks) & (k < n - 1)
    print(f"  {c:2d}   {s05[:, sel].mean():8.0f}  {bar[:, sel].mean():8.0f}  {s67[:, sel].mean():8.0f}  {ctl[:, sel[:-1]].mean():8.0f}  {per[:, sel[:-1]].mean():8.0f}")
reg = st[:, chunks::chunks, 0]
rp = (reg[:, 1:] - reg[:, :-1]).ravel()
print(f"region period mean {rp.mean():.0f} (ideal {4096 * chunks})")
