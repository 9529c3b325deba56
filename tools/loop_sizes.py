"""Static sizes of the loops of a kernel in a variant's ISA (VALU / v_mov / SALU / scalar fetches per loop body, nesting depth):
python tools/loop_sizes.py build/var/<name>/rtx_api-hip-amdgcn-amd-amdhsa-gfx950.s [mangled kernel name] [min VALU]"""
import re, sys
path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "_Z14rtxPass1KernelILb0ELb1ELb1ELi1ELb1EEvN4rtxd6ParamsE"
least = int(sys.argv[3]) if len(sys.argv) > 3 else 40
L, on = [], False
for l in open(path):
    if l.startswith(kern + ":"): on = True
    if on:
        L.append(l.rstrip("\n"))
        if "s_endpgm" in l: break
hdrs = []
for i, l in enumerate(L):
    m = re.search(r"This (Inner )?Loop Header: Depth=(\d+)", l)
    if m:
        j = i
        while not L[j].startswith(".LBB"): j -= 1
        hdrs.append((L[j].split(":")[0], int(m.group(2)), j))
for name, depth, j in hdrs:
    last = None
    for i in range(j, len(L)):
        if re.search(r"s_c?branch\S*\s+" + re.escape(name) + r"\s*$", L[i].split(";")[0].rstrip()): last = i
    if last is None: continue
    body = [l.strip() for l in L[j:last + 1] if l.startswith("\t") and not l.strip().startswith(";")]
    valu = [l for l in body if l.startswith("v_")]
    if len(valu) < least: continue
    idiom = sum(1 for i, l in enumerate(body) if re.match(r"v_cndmask_b32_e64 v\d+, 0, 1, ", l) and any(re.match(r"v_cmp_ne_u32\S* .*0, v\d+", b) for b in body[i + 1:i + 4]))
    print("%-12s depth %d: VALU %4d (v_mov %3d, cndmask+cmp ballots %2d)  SALU %4d  s_load %2d  LDS %3d  VMEM %3d  scratch %2d" % (
        name, depth, len(valu), sum(l.startswith("v_mov") for l in valu), idiom, sum(l.startswith("s_") and not l.startswith("s_load") and not l.startswith("s_waitcnt") and not l.startswith("s_nop") for l in body),
        sum(l.startswith("s_load") for l in body), sum(l.startswith("ds_") for l in body), sum(l.startswith("global_") or l.startswith("buffer_") for l in body), sum(l.startswith("scratch_") for l in body)))
