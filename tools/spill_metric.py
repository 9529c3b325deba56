"""Static register-spill metric of a ray kernel from the compiler's ISA (build/*.s or a variant's .s): v_writelane / v_readlane (SGPR spills to VGPR lanes
cost a VALU issue slot each), scratch loads / stores and v_mov per loop nest level, with the node loop of the wide walk (the loop holding two
s_load_dwordx16 and the prune-record loads) and everything nested in the per-mesh bundle loop counted apart from the per-round code around them.
python tools/spill_metric.py [file.s] [kernel mangled name]"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "build/rtx_api-hip-amdgcn-amd-amdhsa-gfx950.s"
K = sys.argv[2] if len(sys.argv) > 2 else "_Z14rtxPass1KernelILb0ELb1ELb1ELi1ELb1EEvN4rtxd6ParamsE"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(K + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
cur, depth = "top", 0
stat = {}
order = []
for ln in body:
    m = re.search(r"in Loop: Header=(\S+) Depth=(\d+)", ln) or re.search(r"^(\.LBB\S+):.*Loop Header: Depth=(\d+)", ln)
    if m:
        cur = m.group(1).rstrip(":").lstrip("."); depth = int(m.group(2))
    t = ln.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    op = t.split()[0]
    d = stat.setdefault(cur, {"depth": depth, "valu": 0, "wl": 0, "rl": 0, "rfl": 0, "scr": 0, "mov": 0, "salu": 0, "f64": 0, "dpp": 0, "smem": 0})
    if cur not in order:
        order.append(cur)
    if op.startswith("v_"):
        d["valu"] += 1
        if op.startswith("v_writelane"): d["wl"] += 1
        elif op.startswith("v_readlane"): d["rl"] += 1
        elif op.startswith("v_readfirstlane"): d["rfl"] += 1
        elif op.startswith(("v_mov", "v_accvgpr")): d["mov"] += 1
        if "f64" in op: d["f64"] += 1
        if "row_" in t or "quad_perm" in t: d["dpp"] += 1
    elif op.startswith("scratch_"): d["scr"] += 1
    elif op.startswith("s_load"): d["smem"] += 1
    elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")): d["salu"] += 1
tot = {k: sum(s[k] for s in stat.values()) for k in ("valu", "wl", "rl", "rfl", "scr", "mov", "salu", "f64", "dpp")}
print("%-12s %5s %5s %4s %4s %4s %4s %4s %5s %4s %4s" % ("loop", "depth", "valu", "wl", "rl", "rfl", "scr", "mov", "salu", "f64", "dpp"))
for k in order:
    s = stat[k]
    print("%-12s %5d %5d %4d %4d %4d %4d %4d %5d %4d %4d" % (k, s["depth"], s["valu"], s["wl"], s["rl"], s["rfl"], s["scr"], s["mov"], s["salu"], s["f64"], s["dpp"]))
print("%-12s %5s %5d %4d %4d %4d %4d %4d %5d %4d %4d" % ("total", "", tot["valu"], tot["wl"], tot["rl"], tot["rfl"], tot["scr"], tot["mov"], tot["salu"], tot["f64"], tot["dpp"]))
