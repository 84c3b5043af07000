"""One-character-per-instruction trace of a kernel (M mfma, v valu, d lds read, D lds write, g vmem, w waitcnt, | barrier, s salu, B branch):
isa_trace.py file.s kernel-substring"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % key, l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
out = []
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((";", ".")): continue
    if t.endswith(":"): out.append("\n" + t + "\n"); continue
    op = t.split()[0]
    if op.startswith("v_mfma"): c = "M"
    elif op.startswith("v_"): c = "v"
    elif op.startswith("ds_read") or op.startswith("ds_load"): c = "d"
    elif op.startswith("ds_"): c = "D"
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c = "S" if op.startswith("scratch_") else "g"
    elif op.startswith("s_waitcnt"): c = "w"
    elif op.startswith("s_barrier"): c = "|\n"
    elif op.startswith(("s_cbranch", "s_branch")): c = "B"
    elif op.startswith("s_nop"): c = "n"
    else: c = "s"
    out.append(c)
print("".join(out))
