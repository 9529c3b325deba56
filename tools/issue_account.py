"""A DYNAMIC issue account of the pass-1 kernel: static VALU instruction counts of the walk's inner loops
(from the compiler's ISA, build/*.s) x how often each loop body runs in one launch of the headline frame (RTX_DBG wave-level
counters, gpurun_out/r04/r04_dbg_counts.txt), against the SQ_INSTS_VALU the hardware counted for the same launch
(profiles/r04_pass1_pmc.json).  The static count of a loop body is an upper bound of what one trip issues (not every branch
of the body is taken), so the walk's share is an upper bound and "everything else" a lower bound.
python tools/issue_account.py > profiles/r04_issue_account.txt   (also writes profiles/r04_issue_account.json: useful_valu_frac for bench.py)"""
import json, os, re, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"      # python tools/issue_account.py <round tag> > profiles/<tag>_issue_account.txt
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from srchash import source_hash
from isa_mix import klass, issue_cycles

K = "_Z14rtxPass1KernelILb0ELb1ELb1ELi1ELb1EEvN4rtxd6ParamsE"
lines = open(os.path.join(ROOT, "build", "rtx_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(K + ":"))
loops, order, cur = {}, [], "top"
for ln in lines[start + 1:]:
    t = ln.strip()
    if t.startswith("s_endpgm"):
        break
    m = re.search(r"in Loop: Header=(\S+) Depth=(\d+)", ln) or re.search(r"^(\.LBB\S+):.*Loop Header: Depth=(\d+)", ln)
    if m:
        cur = m.group(1).rstrip(":").lstrip(".") + " depth " + m.group(2)
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    op = t.split()[0]
    d = loops.setdefault(cur, {"valu": 0, "cycles": 0.0, "salu": 0, "smem16": 0, "bperm": 0, "vmem": 0, "lds": 0, "readlane": 0, "mov": 0})
    if cur not in order:
        order.append(cur)
    k = klass(op, t)
    if k.startswith("valu_"):
        d["valu"] += 1; d["cycles"] += issue_cycles(op, t)
        if op.startswith(("v_readlane", "v_writelane")): d["readlane"] += 1
        if k == "valu_mov": d["mov"] += 1
    elif k == "salu": d["salu"] += 1
    elif k == "vmem": d["vmem"] += 1
    elif k == "lds": d["lds"] += 1
    if op.startswith("s_load_dwordx16"): d["smem16"] += 1
    if op.startswith("ds_bpermute"): d["bperm"] += 1
# the wide walk: its node loop holds the two s_load_dwordx16 of a WideNode and the two prune-record loads; the pass bodies are the
# loops with the ds_bpermute broadcasts of a survivor's record (first four in program order after the node loop = wide walk: two
# passes per fetch, each compiled twice); the exact test is their innermost child
# (the INNERMOST such loop: since round 6 the object loop around the walk holds a mesh's three record loads and the split constants' reload as well)
node = max((k for k in order if loops[k]["smem16"] >= 2 and loops[k]["vmem"] >= 2), key=lambda k: (loops[k]["smem16"], int(k.split()[-1])))
after = order[order.index(node) + 1:]
depth = lambda k: int(k.split()[-1])
bp = [k for k in after if loops[k]["bperm"] >= 10][:4]
if depth(bp[0]) == depth(node):
    # (until the control-flow campaign of round 6 the broadcasts were counted to the pass loop, in front of the rotated exact-test loop)
    passes = bp
    ex = [k for k in after if depth(k) == depth(passes[0]) + 1 and loops[k]["valu"] > 40][:4]
else:
    # the loops holding the broadcasts ARE the exact-test loops; a pass body is the loop around each of them
    ex = bp
    passes = []
    for e in ex:
        before = order[:order.index(e)]
        passes.append(next(k for k in reversed(before) if depth(k) == depth(e) - 1 and loops[k]["valu"] > 40))
dbg = open(os.path.join(ROOT, "gpurun_out", TAG, TAG + "_dbg_counts.txt")).read()
sec = dbg.split("== nosrc")[0]
m = re.search(r"node visits (\d+), reached leaves (\d+), filter passes \(64 references\) (\d+), of which rejected whole by stage 1 (\d+), by stage 2 (\d+); survivors tested exactly (\d+)", sec)
visits, leaves, npass, rej1, rej2, exact = [int(x) for x in m.groups()]
pmc = json.load(open(os.path.join(ROOT, "profiles", TAG + "_pass1_pmc.json")))
kern = [v for n, v in pmc["workloads"]["headline"]["kernels"].items() if "Pass1Kernel<false, true, true" in n][0]
total = kern["SQ_INSTS_VALU"]
pv = sum(loops[k]["valu"] for k in passes) / len(passes)
ev = sum(loops[k]["valu"] for k in ex) / max(len(ex), 1)
print("sources %s, counters of sources %s; headline frame, rtxPass1Kernel<false, true, true>, one launch" % (source_hash(), pmc["source_hash"]))
print("SQ_INSTS_VALU (hardware)                                   %.3e wave-instructions" % total)
rows = [("node visit (pop, WideNode + prune records, pruneEval8 on 48 lanes, 8 slot tests, pushes)", node, loops[node]["valu"], visits),
        ("filter pass of 64 references (assign, bundleRejects1/2, ballots; without the exact tests)", passes[0], pv, npass),
        ("exact test of one survivor (record through ds_bpermute, Moller-Trumbore)", ex[0] if ex else "-", ev, exact)]
acc = 0
for name, k, v, n in rows:
    acc += v * n
    print("%-92s static %4d VALU x %9d trips = %.3e  (<= %4.1f %%)   [%s: readlane/writelane %d, v_mov %d, SALU %d, LDS %d, VMEM %d]" % (
        name, v, n, v * n, 100.0 * v * n / total, k, loops.get(k, {}).get("readlane", 0), loops.get(k, {}).get("mov", 0), loops.get(k, {}).get("salu", 0), loops.get(k, {}).get("lds", 0), loops.get(k, {}).get("vmem", 0)))
print("sum of the three static upper bounds %.3e = %.0f %% of the measured count: the bodies are not executed in full (a pass that stage 1 rejects whole -- %d of %d -- skips stage 2; %d more end after stage 2)" % (acc, 100.0 * acc / total, rej1, npass, rej2))
salu = kern.get("SQ_INSTS_SALU", 0)
print("scalar side of the same loops (SQ_INSTS_SALU %.3e per launch = %.2f per VALU instruction): node visit %d SALU, filter pass %d, exact test %d (static)" % (
    salu, salu / total, loops[node]["salu"], loops[passes[0]]["salu"], loops[ex[0]]["salu"] if ex else 0))
m2 = re.search(r"per-ray slot tests (\d+)", sec)
# USEFUL arithmetic = what the reference's semantics need of this kernel: the per-ray box tests of the slots a walk really tests (21 VALU each: 6 sub,
# 6 mul, 3 min, 3 max, max3, min3, compare) and the exact Moller-Trumbore tests of the survivors (the static body); everything else -- prune evaluation,
# bundle filter, stack, state machine, parking -- is the machinery that decides what NOT to test
slot_tests = int(m2.group(1)) if m2 else None
useful = (21.0 * slot_tests if slot_tests else 0.0) + ev * exact
print("useful arithmetic (per-ray slot box tests %s x 21 + exact tests %d x %d static): %.3e = %.1f %% of the issued VALU instructions" % (slot_tests, exact, ev, useful, 100.0 * useful / total))
json.dump({"source_hash": pmc["source_hash"], "useful_valu_basis": "(per-ray slot box tests x 21 VALU + exact tests x the static body of the Moller-Trumbore loop) / SQ_INSTS_VALU; counts: RTX_DBG build of the same sources",
           "workloads": {"headline": {"useful_valu_frac": round(useful / total, 4), "node_visits": visits, "reached_leaves": leaves, "filter_passes": npass, "exact_tests": exact, "slot_tests": slot_tests}}},
          open(os.path.join(ROOT, "profiles", TAG + "_issue_account.json"), "w"), indent=1)
print("spill traffic inside the loops above: v_readlane / v_writelane %d in the node loop, %d per pass body, %d per exact test (VERDICT r2: 122 SGPR spill slots in the loops -- the spills that remain sit in the round loop around the walk)" % (
    loops[node]["readlane"], loops[passes[0]]["readlane"], loops[ex[0]]["readlane"] if ex else 0))
