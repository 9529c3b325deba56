"""Static instruction counts of a kernel's ISA between consecutive loop headers (a coarse map of where the VALU instructions, copies and spills sit):
python tools/region_sizes.py build/var/<name>/rtx_api-hip-amdgcn-amd-amdhsa-gfx950.s [mangled kernel name]"""
import re, sys
path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "_Z14rtxPass1KernelILb0ELb1ELb1ELi1ELb1EEvN4rtxd6ParamsE"
L, on = [], False
for l in open(path):
    if l.startswith(kern + ":"): on = True
    if on:
        L.append(l.rstrip("\n"))
        if "s_endpgm" in l: break
marks = [(0, "entry", 0)]
for i, l in enumerate(L):
    m = re.search(r"This (Inner )?Loop Header: Depth=(\d+)", l)
    if m:
        j = i
        while not L[j].startswith(".LBB"): j -= 1
        marks.append((j, L[j].split(":")[0], int(m.group(2))))
marks.append((len(L), "end", 0))
for (a, name, d), (b, _, _) in zip(marks, marks[1:]):
    body = [l.strip() for l in L[a:b] if l.startswith("\t") and not l.strip().startswith(";")]
    v = [l for l in body if l.startswith("v_")]
    print("%-12s depth %d  lines %5d-%5d  VALU %4d  v_mov %3d  lane r/w %3d  cndmask %3d  SALU %4d  scratch %2d  LDS %3d  VMEM %3d" % (
        name, d, a, b, len(v), sum(l.startswith("v_mov") for l in v), sum(l.startswith("v_readlane") or l.startswith("v_writelane") for l in v), sum(l.startswith("v_cndmask") for l in v),
        sum(l.startswith("s_") and not l.startswith("s_waitcnt") and not l.startswith("s_nop") for l in body), sum(l.startswith("scratch") for l in body), sum(l.startswith("ds_") for l in body),
        sum(l.startswith("global_") for l in body)))
