"""Instruction-class breakdown of the pass-1 kernel from the compiler's ISA (build/*.s, written by build.sh):
whole kernel and per innermost loop.  python tools/isa_mix.py [round tag] -> profiles/<tag>_pass1_isa.json
Classes: packed fp32 arithmetic (v_pk_*), plain arithmetic (mul/add/fma/max/min/...), compare + select, cross-lane
(readlane / writelane / DPP / mbcnt), moves, VMEM, SMEM, SALU, other."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from srchash import source_hash, ROOT

# (round 6: every mesh kernel exists per culling mode -- Li1 / Li0 -- and per PLAIN -- Lb1: scenes of Diffuse objects under point / distant lights, what the headline runs)
KERNELS = {"rtxPass1Kernel<false, true, true>": "_Z14rtxPass1KernelILb0ELb1ELb1ELi1ELb1EEvN4rtxd6ParamsE", "rtxFrameKernel<true, true>": "_Z14rtxFrameKernelILb1ELb1ELi1ELb1EEvN4rtxd6ParamsE",
           "rtxSsaaKernel<false, true, true>": "_Z13rtxSsaaKernelILb0ELb1ELb1ELi1ELb1EEvN4rtxd6ParamsE",
           "rtxPass1Kernel<false, true, true, 1, false> (any material)": "_Z14rtxPass1KernelILb0ELb1ELb1ELi1ELb0EEvN4rtxd6ParamsE",
           "rtxPass1Kernel<false, true, true, 0, true> (culling off)": "_Z14rtxPass1KernelILb0ELb1ELb1ELi0ELb1EEvN4rtxd6ParamsE"}


def klass(op, line):
    if op.startswith("v_pk_"):
        return "valu_packed_arith"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "valu_cmp_select"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_mbcnt")) or "row_" in line or "quad_perm" in line:
        return "valu_cross_lane"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "valu_mov"
    if op.startswith("v_"):
        return "valu_plain_arith"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


# Issue cost of a VALU instruction on gfx950 (tools/ubench/valu_rate.hip, profiles/r02_valu_rate.txt): fp32 add / sub /
# mul / fma / fmac, moves and simple integer / logic operations whose sources are VGPRs or inline constants issue in ~2.4
# cycles per wave64 instruction per SIMD; everything else -- any instruction with an SGPR (or VCC / literal) source, min / max /
# max3 / med3, compares, selects, shifts, integer multiplies, v_readlane, DPP, packed and fp64 arithmetic -- in ~4.2; the
# transcendental unit (rcp, rsq, sqrt, ...) in ~8.2.
FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def issue_cycles(op, t):
    base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
    if base.startswith(TRANS):
        return 8.2
    operands = t[len(op):]
    scalar_src = bool(re.search(r"(?<![a-z0-9_\[])(s\d+|s\[\d+:\d+\]|vcc|exec|m0|0x[0-9a-f]+)\b", operands.split(",", 1)[1] if "," in operands else ""))
    if base in FAST and not scalar_src and "row_" not in t and "quad_perm" not in t:
        return 2.4
    return 4.2


def one(path, lines, name, KERNEL):
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    total = {}
    loops = {}
    cyc = {}
    cur = ("top", 0)
    for ln in lines[start + 1:]:
        t = ln.strip()
        if t.startswith("s_endpgm"):
            break
        m = re.search(r"in Loop: Header=(\S+) Depth=(\d+)", ln) or re.search(r"^(\.LBB\S+):.*Loop Header: Depth=(\d+)", ln)
        if m:
            cur = (m.group(1).rstrip(":").lstrip("."), int(m.group(2)))
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        k = klass(op, t)
        total[k] = total.get(k, 0) + 1
        d = loops.setdefault("%s (depth %d)" % cur, {})
        d[k] = d.get(k, 0) + 1
        if k.startswith("valu_"):
            c = issue_cycles(op, t)
            cyc[c] = cyc.get(c, 0) + 1
    valu = sum(v for k, v in total.items() if k.startswith("valu_"))
    res = {"kernel": name, "kernel_total": total, "valu_total": valu,
           "valu_by_issue_cycles": {str(k): v for k, v in sorted(cyc.items())},
           "valu_issue_cycles_static_mean": round(sum(k * v for k, v in cyc.items()) / max(valu, 1), 3),
           "valu_fraction_by_class": {k: round(v / valu, 4) for k, v in total.items() if k.startswith("valu_")},
           "by_innermost_loop": {k: v for k, v in sorted(loops.items(), key=lambda kv: -sum(kv[1].values()))[:12]}}
    txt = "\n".join(lines)
    i = txt.index(".name:           " + KERNEL)
    blk = txt[i:i + 1500]      # (the keys wanted follow .name in the kernel's metadata map; looking in front of it finds the previous kernel's)
    for key in ("sgpr_count", "vgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size"):
        m = re.search(r"\.%s:\s+(\d+)" % key, blk)
        if m:
            res[key] = int(m.group(1))
    return res


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    path = os.path.join(ROOT, "build", "rtx_api-hip-amdgcn-amd-amdhsa-gfx950.s")
    lines = open(path).read().split("\n")
    per = {name: one(path, lines, name, sym) for name, sym in KERNELS.items()}
    # top level = the pass-1 kernel (what bench.py quotes beside the headline roofline); "kernels" holds all three
    res = {"source_hash": source_hash(), "what": "static instruction counts from " + os.path.basename(path)}
    res.update(per["rtxPass1Kernel<false, true, true>"])
    res["kernels"] = per
    json.dump(res, open(os.path.join(ROOT, "profiles", "%s_pass1_isa.json" % tag), "w"), indent=1)
    for name, r in per.items():
        print(name, {k: r[k] for k in r if k not in ("by_innermost_loop", "kernel")})


if __name__ == "__main__":
    main()
