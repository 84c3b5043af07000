"""Register/scratch/LDS footprint of every kernel in a hipcc -S listing: isa_meta.py file.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if key not in name: continue
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
    print(f"{name[:70]:70s} vgpr={g('vgpr_count'):>4s} agpr={blk.split()[0]:>4s} sgpr={g('sgpr_count'):>4s} scratch={g('private_segment_fixed_size'):>5s}")
