"""Instruction-class trace of one kernel's barrier-to-barrier regions in a hipcc -S listing:
   isa_trace.py file.s <mangled-name substring>     (M mfma, r/w LDS read/write, G global load, S global store, v VALU, s SALU, | waitcnt, B barrier)"""
import sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
def cls(l):
    l = l.strip()
    for pre, c in (("v_mfma", "M"), ("ds_read", "r"), ("ds_load", "r"), ("ds_write", "w"), ("ds_store", "w"), ("global_load", "G"), ("buffer_load", "G"),
                   ("global_store", "S"), ("s_waitcnt", "|"), ("s_barrier", "B"), ("s_cbranch", "j"), ("v_", "v"), ("s_", "s")):
        if l.startswith(pre): return c
    return "\n" + l + "\n" if l.startswith(".LBB") else ""
print("".join(cls(l) for l in lines[start:end + 1]))
