"""Instruction mix per basic block of one kernel in a hipcc -save-temps .s file:
    python scripts/isa_mix.py file.s KERNEL_SYMBOL_SUBSTRING [min_instrs]"""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
src = open(path).read().split("\n")
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\S*:", l) and key in l.split(":")[0])
end = next(i for i in range(start, len(src)) if ".end_amdhsa_kernel" in src[i])
lines = []
for l in src[start + 1:end]:
    t = l.split(";")[0].strip()
    if not t or (t.startswith(".") and not t.endswith(":")):
        continue
    lines.append(t)


def classify(l):
    op = l.split()[0]
    if op.endswith(":"):
        return "label"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "vld"
    if op.startswith(("global_store", "buffer_store", "flat_store")):
        return "vst"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


seg, name, first = Counter(), "entry", 0
tot = Counter()
for i, l in enumerate(lines + ["END:"]):
    c = classify(l)
    if c == "label":
        if sum(seg.values()) >= min_n:
            print(f"{name:14s} @{first:5d} n={sum(seg.values()):4d}  " + " ".join(f"{k}={v}" for k, v in sorted(seg.items())))
        tot.update(seg)
        seg, name, first = Counter(), l, i
    else:
        seg[c] += 1
print("total", dict(tot))
for l in src[end:end + 400]:
    if re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|SGPRBlocks|NumSgprs)", l):
        print(l.strip())
    if "codeLenInByte" in l:
        print(l.strip()); break
