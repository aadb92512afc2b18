#!/usr/bin/env python3
"""Diagnostic: where a kernel's scratch (spill) traffic sits in the final assembly, and what the reloaded values feed.
usage: scratch_map.py <device .s> <substring of the kernel's mangled name> [lo hi]"""
import re, sys, collections
asm, key = sys.argv[1], sys.argv[2]
lines, on = [], False
for l in open(asm):
    if re.match(r"^_Z\w*:", l) and key in l:
        on = True
    if on:
        lines.append(l.rstrip("\n"))
        if l.strip().startswith(".Lfunc_end"):
            break
print(len(lines), "lines")
B = 500
for i in range(0, len(lines), B):
    seg = lines[i:i + B]
    c = lambda k: sum(k in l for l in seg)
    print(f"{i:6d} mfma {c('v_mfma'):4d} scratch ld {c('scratch_load'):3d} st {c('scratch_store'):3d} rsq {c('v_rsq_f32'):2d} vmem {c('global_load') + c('buffer_load'):3d} ds {c('ds_read') + c('ds_write'):3d}")
if len(sys.argv) > 4:
    lo, hi = int(sys.argv[3]), int(sys.argv[4])
    uses = collections.Counter()
    for i in range(lo, min(hi, len(lines))):
        if "scratch_load" in lines[i]:
            m = re.search(r"scratch_load_dword\w*\s+v\[?(\d+)", lines[i])
            off = re.search(r"offset:(\d+)", lines[i])
            use = None
            for j in range(i + 1, min(i + 40, len(lines))):
                if re.search(r"\bv\[?%s\b" % m.group(1), lines[j]) and "scratch_load" not in lines[j]:
                    use = lines[j].split()[0]
                    break
            uses[(off.group(1) if off else "0", use)] += 1
    for (o, u), n in uses.most_common(40):
        print(f"   offset {o:>5} x{n:3d} -> {u}")
