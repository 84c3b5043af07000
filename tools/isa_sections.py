"""Instruction mix of one kernel in a hipcc -S listing, split at s_memtime markers (SAB timeline build) or labels.
usage: isa_sections.py file.s kernel-substring"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % key, l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
sec, cnt, ops = 0, collections.Counter(), collections.Counter()
def flush():
    print(f"section {sec:2d}: " + "  ".join(f"{k}={v}" for k, v in sorted(cnt.items())) + "   top valu: " + ", ".join(f"{k}:{v}" for k, v in ops.most_common(6)))
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"): continue
    op = t.split()[0]
    if op == "s_memtime":
        flush(); sec += 1; cnt, ops = collections.Counter(), collections.Counter(); continue
    cnt[kind(op)] += 1
    if kind(op) == "valu": ops[op] += 1
flush()
