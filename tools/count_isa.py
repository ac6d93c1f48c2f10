#!/usr/bin/env python3
"""Count the fp64 VALU instructions the main kernels execute per integrand evaluation, from the gfx950 ISA hipcc
emits for is3d_amd/csrc/cf_kernels.hip, cf_feqmod.hip and cf_vah.hip, and write is3d_amd/csrc/isa_counts.json (read by bench.py for the
fp64-VALU roofline).  Method: for every cf_main_* instantiation take the basic blocks of the innermost loop
that contains v_rcp_f64 (exactly one v_rcp_f64 is issued per evaluation) and divide the opcode histogram of
those blocks by the number of evaluations they cover.  That number is READ FROM THE ISA: every evaluation ends in exactly one
accumulate `acc = fma(p.dsigma, w, acc)`, and the accumulators are the registers of the loop that nothing but such an in-place
FMA ever writes (v_fmac_f64 D, A, B or v_fma_f64 D, A, B, D: destination == addend; the reload of a spilled accumulator does not count as a
write) -- temporaries that also happen to be updated in place (Newton steps, Horner chains) are initialised by some other instruction inside
the loop and drop out.  The reciprocal batch
evals_per_rcp = evaluations / v_rcp_f64 follows (4 for the 3+1D 8 x 7 tiles, 8 for the 14-moment 2+1D 8-wide tiles, 4 for the
Chapman-Enskog 8 x 31 tile, ...); the template rule the kernels use is kept beside it as evals_per_rcp_template, and the two are
compared by tests/test_isa_counts.py.
For the 3+1D tile kernel that loop is the whole unit (header + rows),
so the amortised exponentials are included; for 2+1D the per-unit header (<2 %) is outside the counted loop.

flops: v_fma/v_fmac = 2, every other fp64 VALU op = 1 (v_rcp_f64 counts 1 although it issues at 1/4 rate).
issue_cycles: 4 cycles per fp64-rate VALU wave-instruction, 16 for v_rcp_f64 / v_rsq_f64 (tools/ubench_fp64.hip measures
4.36-4.57 and 16.3-16.5 in sustained runs at the probed shader clock), 2 for 32-bit integer VALU ops.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "is3d_amd", "csrc", f) for f in ("cf_kernels.hip", "cf_feqmod.hip", "cf_vah.hip")]
AUDIT_ONLY = [os.path.join(ROOT, "is3d_amd", "csrc", f) for f in ("cf_yield.hip", "cf_sampler.hip", "cf_multi.hip")]
OUT = os.path.join(ROOT, "is3d_amd", "csrc", "isa_counts.json")

F64_OPS = ["v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_rcp_f64", "v_ldexp_f64",
           "v_rndne_f64", "v_cvt_i32_f64", "v_mov_b64", "v_cmp", "v_cndmask_b32"]


def demangle_params(sym):
    m = re.match(r"_ZN4is3d(\d+)(cf_main_[a-z0-9]+)I(.*?)EEvPKd", sym)
    if not m:
        return None
    name, args = m.group(2), m.group(3)
    vals = re.findall(r"L([bi])(\d+)E", args)
    vals = [int(v) for _, v in vals]
    if name == "cf_main_tile":
        keys = ["CE", "DIM3", "OUTFLOW", "REG", "BARYON", "JT", "R", "LAZY", "DMA"]
    elif name == "cf_main_tile3e":
        keys = ["CE", "OUTFLOW", "REG", "JT", "R", "MODE", "PROF", "BARYON", "E2G", "RAWH", "E2L"]
    elif name == "cf_main_tile3s":
        keys = ["CE", "OUTFLOW", "REG", "JT", "R"]
    elif name == "cf_main_feqmod":
        keys = ["DIM3", "OUTFLOW", "MODE3", "JT", "R", "BARYON", "ROWS", "PROF"]
    elif name == "cf_main_vah":
        keys = ["DIM3", "REG", "JT", "R", "LDSD"]
    elif name == "cf_main_vah3":
        keys = ["DIM3", "REG", "JT", "R", "LDSD"]
    else:
        keys = ["CE", "DIM3", "OUTFLOW", "REG", "KT"]
    return name, dict(zip(keys, vals))


def vregs(tok):
    """VGPR numbers an operand names: v12 -> [12], v[10:11] -> [10, 11], anything else -> []."""
    tok = tok.strip().lstrip("-").strip("|")
    m = re.match(r"^v(\d+)$", tok)
    if m:
        return [int(m.group(1))]
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def count_evaluations(ins):
    """Accumulate-FMAs of the loop body `ins` ([(mnemonic, operand text)]): in-place FMAs (destination == addend) into registers that no
    other instruction of the loop writes."""
    writers = collections.defaultdict(set)     # vgpr -> set of instruction indices that write it
    cand = {}                                  # instruction index -> tuple of its destination registers
    for i, (op, args) in enumerate(ins):
        ops = [a.strip() for a in args.split(",")]
        if not ops or op.startswith(("ds_write", "global_store", "scratch_store", "buffer_store", "s_", "v_cmp", "v_readfirstlane", "v_readlane")):
            continue
        if op.startswith("global_load_lds"):
            continue
        dst = vregs(ops[0])
        if op.startswith(("v_", "ds_read", "global_load", "scratch_load", "buffer_load")):
            for r in dst:
                writers[r].add(i)
        if op == "v_fmac_f64" and len(dst) == 2:
            cand[i] = tuple(dst)
        elif op == "v_fma_f64" and len(ops) >= 4 and len(dst) == 2:
            add = ops[3].split()[0]            # drop a trailing modifier ("clamp", "mul:2")
            if vregs(add) == dst and not ops[3].strip().startswith("-") and "clamp" not in args:
                cand[i] = tuple(dst)
    # a spilled accumulator comes back through a reload (scratch_load / v_accvgpr_read into the same registers): that is not an initialisation
    reload = {i for i, (op, _) in enumerate(ins) if op.startswith(("scratch_load", "v_accvgpr_read"))}
    n = 0
    counted = set()
    for i, dst in cand.items():
        if all(all((j in cand and cand[j] == dst) or j in reload for j in writers[r]) for r in dst):
            n += 1
            counted.add(i)
    # an accumulator that LIVES in a spill slot: reload into a scratch register that other code also uses, accumulate, store back to the same slot
    def slot(args):
        m = re.search(r"offset:(\d+)", args)
        return int(m.group(1)) if m else 0
    for i, dst in cand.items():
        if i in counted:
            continue
        prev = max((j for r in dst for j in writers[r] if j < i), default=None)
        if prev is None or not ins[prev][0].startswith("scratch_load") or tuple(vregs(ins[prev][1].split(",")[0])) != dst:
            continue
        for j in range(i + 1, len(ins)):
            op, args = ins[j]
            if op.startswith("scratch_store") and tuple(vregs(args.split(",")[1])) == dst:
                if slot(args) == slot(ins[prev][1]):
                    n += 1
                break
            if any(j in writers[r] for r in dst):
                break
    return n


def main():
    text = []
    flat_main = {}
    with tempfile.TemporaryDirectory() as td:
        for src in SRCS:
            s_path = os.path.join(td, "k.s")
            # -DIS3D_DEV: the developer build holds every kernel instantiation -- the ones the shipped library launches (identical code: the switch only
            # gates launchers) and the A/B variants of rounds 1-5 that tools/gpu_ab.py and `IS3D_USE_DEV_LIB=1 bench.py --variant N` still price
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DIS3D_DEV", "-S", "--cuda-device-only",
                                   "-I", os.path.dirname(src), src, "-o", s_path], stderr=subprocess.DEVNULL)
            asm = open(s_path).read()
            text += asm.split("\n")
            flat_main[os.path.basename(src)] = len(re.findall(r"^\s+flat_(?:load|store|atomic)", asm, re.M))
        # address-space audit over EVERY kernel source: a flat_* instruction means a pointer lost its address space (an LDS pointer rounded
        # through uintptr_t did, in cf_prep: its reads became flat_load + s_waitcnt vmcnt(0), which on gfx9 also waits for every store in flight)
        flat = dict(flat_main)
        for src in AUDIT_ONLY:
            s_path = os.path.join(td, "a.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                                   "-I", os.path.dirname(src), "-I", os.path.join(ROOT, "include"), src, "-o", s_path], stderr=subprocess.DEVNULL)
            flat[os.path.basename(src)] = len(re.findall(r"^\s+flat_(?:load|store|atomic)", open(s_path).read(), re.M))
    kernels = {}
    resources = collections.defaultdict(dict)   # symbol -> {NumVgprs, NumAgprs, ScratchSize, Occupancy, ...} from the compiler's resource comments
    cur, blocks, meta_for = None, None, None
    for ln in text:
        if cur is None and meta_for is not None:
            m = re.match(r"^; (NumVgprs|NumAgprs|TotalNumSgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", ln)
            if m:
                resources[meta_for].setdefault(m.group(1), int(m.group(2)))
        m = re.match(r"^(_ZN4is3d\d+cf_main_\w+):", ln)
        if m:
            cur = m.group(1)
            blocks = [dict(depth=0, ops=collections.Counter())]
            kernels[cur] = blocks
            continue
        if cur is None:
            continue
        if ln.startswith(".Lfunc_end"):   # (an early-exit s_endpgm can sit in the middle of the body)
            meta_for = cur                # the resource comments ("; NumVgprs: 256", "; ScratchSize: 8") follow the function body
            cur = None
            continue
        m = re.match(r"^\.LBB\d+_\d+:(.*)", ln)
        if m:
            d = re.search(r"Depth=(\d+)", m.group(1))
            blocks.append(dict(depth=int(d.group(1)) if d else 0, ops=collections.Counter()))
            continue
        m = re.match(r"^\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|scratch_[a-z0-9_]+|buffer_[a-z0-9_]+)\s*(.*)", ln)
        if m:
            blocks[-1]["ops"][m.group(1).replace("_e32", "").replace("_e64", "")] += 1
            blocks[-1].setdefault("ins", []).append((m.group(1).replace("_e32", "").replace("_e64", ""), m.group(2)))
    out = {}
    for sym, blocks in kernels.items():
        p = demangle_params(sym)
        if not p:
            continue
        name, params = p
        if params.pop("PROF", 0):   # the cycle-accounting instantiation (dev) is not a product kernel
            continue
        ldsd = params.pop("LDSD", None) if name in ("cf_main_vah3", "cf_main_vah") else None
        # cf_main_vah3 in 3+1D exists twice: LDSD = 1100 is the one-wave workgroup the plan launches by default (cf_vah.hip), 1536 the two-wave
        # one of waves_per_group = 2 -- the plain key is the default's, the other carries its LDSD
        vah3_alt = name == "cf_main_vah3" and params.get("DIM3") == 1 and ldsd == 1536
        if name == "cf_main_tile":
            params.pop("LAZY", None)
            if not params.pop("DMA", 1):   # the register-staged copy kept for A/B (variant 8)
                continue
        # M0 audit (cf_math.h::glds16a writes M0 inside an asm statement the compiler cannot be told about): every instruction of the kernel that
        # names m0 or uses it implicitly, besides the helper's own s_mov_b32 m0 / global_load_lds pair
        all_ins = [x for b in blocks for x in b.get("ins", [])]
        m0_set = sum(1 for op, args in all_ins if op == "s_mov_b32" and args.split(",")[0].strip() == "m0")
        glds = sum(1 for op, _ in all_ins if op.startswith("global_load_lds"))
        implicit = ("v_movrel", "s_movrel", "v_interp", "s_sendmsg", "ds_gws", "ds_read_addtid", "ds_write_addtid", "buffer_load_lds", "s_ttrace")
        m0_other = [op for op, args in all_ins if (re.search(r"\bm0\b", args) and not (op == "s_mov_b32" and args.split(",")[0].strip() == "m0"))
                    or op.startswith(implicit)]
        rb = [b for b in blocks if b["ops"].get("v_rcp_f64", 0) > 0]
        if not rb:
            continue
        dmax = max(b["depth"] for b in rb)
        hot = collections.Counter()
        hot_ins = []
        for b in blocks:
            if b["depth"] >= dmax and dmax > 0:
                hot.update(b["ops"])
                hot_ins += b.get("ins", [])
        if hot["v_rcp_f64"] == 0:
            continue
        if name == "cf_main_direct":
            rbatch_t = 1
        else:
            jt = params["JT"]
            rbatch_t = (4 if jt % 4 == 0 else (3 if jt % 3 == 0 else 2)) if params.get("DIM3", 1) else (8 if jt % 8 == 0 else 4)
        n_eval = count_evaluations(hot_ins)
        from_isa = n_eval > 0 and n_eval % hot["v_rcp_f64"] == 0
        if not from_isa:   # accumulators spilled or renamed inside the loop: fall back to the template rule and say so
            n_eval = hot["v_rcp_f64"] * rbatch_t
        rbatch = n_eval // hot["v_rcp_f64"]
        f64 = {k: v for k, v in hot.items() if k.endswith("_f64") or k == "v_mov_b64"}
        fma = hot["v_fma_f64"] + hot["v_fmac_f64"]
        other = sum(v for k, v in f64.items() if k not in ("v_fma_f64", "v_fmac_f64", "v_mov_b64"))
        flops = 2 * fma + other
        trans = hot["v_rcp_f64"] + hot["v_rsq_f64"]
        full_rate = sum(v for k, v in f64.items()) - trans
        int_ops = sum(v for k, v in hot.items() if k.startswith("v_") and k not in f64)
        cycles = 4 * full_rate + 16 * trans + 2 * int_ops
        if name == "cf_main_feqmod" and params.pop("BARYON", 0):
            name = "cf_main_feqmod_baryon"
        if name == "cf_main_tile3e" and params.pop("E2G", 0):   # the developer build's variant 10 (E2 column from global memory): its own key
            name = "cf_main_tile3e_e2g"
        if name == "cf_main_tile3e" and params.pop("RAWH", 0):   # variant 11 (raw header values as FMA operands)
            name = "cf_main_tile3e_rawh"
        params.pop("RAWH", None)
        if name == "cf_main_tile3e" and params.pop("E2L", 0):    # variant 12 (tables built per workgroup in LDS)
            name = "cf_main_tile3e_e2l"
        params.pop("E2L", None)
        if name in ("cf_main_tile3e", "cf_main_tile3e_e2g", "cf_main_tile3e_rawh", "cf_main_tile3e_e2l") and params.pop("BARYON", 0):
            name += "_baryon"
        key = "%s:%s" % (name, ",".join("%s=%d" % kv for kv in params.items()))
        if vah3_alt:
            key += ",LDSD=1536"
        # spills: scratch instructions of the counted loop (per pass of the loop and per evaluation) and of the whole kernel.  One in the loop sits on
        # the critical path of every unit; one outside (prologue / epilogue: lane constants parked while the accumulators are live) is paid once per wave
        scr_loop = sum(v for k, v in hot.items() if k.startswith("scratch_"))
        scr_all = sum(1 for op, _ in all_ins if op.startswith("scratch_"))
        res = resources.get(sym, {})
        out[key] = dict(scratch_in_loop=scr_loop, scratch_in_kernel=scr_all, scratch_bytes_per_lane=res.get("ScratchSize"), vgprs=res.get("NumVgprs"), sgprs=res.get("TotalNumSgprs"),
                        occupancy_waves_per_simd=res.get("Occupancy"),
                        evals_in_loop=n_eval, evals_per_rcp=rbatch, evals_per_rcp_template=rbatch_t, evals_counted_from_isa=bool(from_isa), m0_writes=m0_set, global_load_lds=glds, m0_other_users=m0_other, flop_per_eval=round(flops / n_eval, 3), valu_f64_instr_per_eval=round(sum(f64.values()) / n_eval, 3),
                        issue_cycles_per_eval=round(cycles / n_eval, 2), lds_instr_per_eval=round(sum(v for k, v in hot.items() if k.startswith("ds_")) / n_eval, 3),
                        histogram={k: v for k, v in sorted(hot.items()) if v and (k.startswith("v_") or k.startswith("ds_"))})
    # ---- the sampler's Gauss-Laguerre density kernel (cf_sampler.hip): fp64 instructions and flops per quadrature node, from its innermost loops
    # (one v_rcp_f64 per node; the compiler versions the loop for the optional J10 / J20 integrals -- the leanest version is the plain n_eq integral
    # that df_mode 1, 2, 4 run)
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(ROOT, "is3d_amd", "csrc", "cf_sampler.hip")
        s_path = os.path.join(td, "s.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-I", os.path.dirname(src), "-I", os.path.join(ROOT, "include"), src, "-o", s_path], stderr=subprocess.DEVNULL)
        loops, cur_loop, inside = [], None, False
        for ln in open(s_path).read().split("\n"):
            if re.match(r"^_ZN4is3d18cf_sampler_density", ln):
                inside = True
                continue
            if not inside:
                continue
            if ln.startswith(".Lfunc_end"):
                break
            m = re.match(r"^\.LBB\d+_\d+:(.*)", ln)
            if m:
                d = re.search(r"Depth=(\d+)", m.group(1))
                if d and int(d.group(1)) >= 1:
                    if cur_loop is None or "Parent Loop" not in m.group(1) and "in Loop: Header" in m.group(1):
                        cur_loop = collections.Counter()
                        loops.append(cur_loop)
                else:
                    cur_loop = None
                continue
            m = re.match(r"^\s+(v_[a-z0-9_]+)\s", ln)
            if m and cur_loop is not None:
                cur_loop[m.group(1).replace("_e32", "").replace("_e64", "")] += 1
        best = None
        for lp in loops:
            nodes = lp.get("v_rcp_f64", 0)
            if not nodes:
                continue
            f64 = {k: v for k, v in lp.items() if k.endswith("_f64")}
            fma = lp.get("v_fma_f64", 0) + lp.get("v_fmac_f64", 0)
            per = dict(nodes_in_loop=nodes, valu_f64_instr_per_node=round(sum(f64.values()) / nodes, 3),
                       flop_per_node=round((2 * fma + sum(v for k, v in f64.items() if k not in ("v_fma_f64", "v_fmac_f64"))) / nodes, 3),
                       histogram={k: v for k, v in sorted(lp.items()) if k.endswith("_f64")})
            if best is None or per["valu_f64_instr_per_node"] < best["valu_f64_instr_per_node"]:
                best = per
        if best:
            out["cf_sampler_density"] = best
    out["_audit"] = dict(flat_instructions=flat)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    if "-v" in sys.argv:
        for k, v in sorted(out.items()):
            if k.startswith("_") or "flop_per_eval" not in v:
                continue
            print(k, v["flop_per_eval"], v["valu_f64_instr_per_eval"], v["issue_cycles_per_eval"])
    print("wrote", OUT, len(out) - 1, "kernels")


if __name__ == "__main__":
    main()
