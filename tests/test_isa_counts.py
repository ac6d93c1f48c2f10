"""CPU: is3d_amd/csrc/isa_counts.json (tools/count_isa.py), the table bench.py's fp64-VALU roofline is priced with.  The number of
evaluations a kernel's hot loop covers is read from the emitted ISA (accumulate-FMAs into registers nothing else writes), not re-derived
from the template rules; this test holds the table to that and to a sanity band."""
import json
import os
import re

from conftest import ROOT


def test_isa_counts_are_read_from_the_isa_and_sane():
    d = json.load(open(os.path.join(ROOT, "is3d_amd", "csrc", "isa_counts.json")))
    audit = d.pop("_audit")
    # no pointer of any kernel lost its address space: a flat_load waits on vmcnt, which on gfx9 also counts the stores in flight
    # (cf_prep's descriptor table once did: every trip of its record writer then waited for all of the wave's outstanding stores)
    assert set(audit["flat_instructions"]) >= {"cf_kernels.hip", "cf_feqmod.hip", "cf_vah.hip", "cf_yield.hip", "cf_sampler.hip", "cf_multi.hip"}
    assert all(n == 0 for n in audit["flat_instructions"].values()), audit["flat_instructions"]
    assert len(d) >= 100
    # the sampler's Gauss-Laguerre density kernel is counted per quadrature node (one v_rcp_f64 and one v_rsq_f64 each)
    dens = d.pop("cf_sampler_density")
    assert dens["histogram"]["v_rcp_f64"] == dens["nodes_in_loop"] == dens["histogram"]["v_rsq_f64"]
    assert 25.0 <= dens["valu_f64_instr_per_node"] <= 50.0 and 40.0 <= dens["flop_per_node"] <= 90.0
    names = set()
    for key, v in d.items():
        assert key.startswith("cf_main_"), key
        names.add(key.split(":")[0])
        if key.startswith(("cf_main_tile3e_e2g", "cf_main_tile3e_e2l")):   # round 5's dropped experiment (developer build): some instantiations keep accumulators in scratch and are counted by the template rule
            continue
        assert v["evals_counted_from_isa"], key
        assert 20.0 <= v["flop_per_eval"] <= 70.0, (key, v["flop_per_eval"])
        assert 13.0 <= v["valu_f64_instr_per_eval"] <= 45.0, (key, v["valu_f64_instr_per_eval"])
        m = re.search(r"JT=(\d+)", key)
        if m:
            assert v["evals_in_loop"] % int(m.group(1)) == 0, key       # evaluations come in whole rows of JT
        assert v["evals_in_loop"] == v["evals_per_rcp"] * v["histogram"]["v_rcp_f64"], key
    assert {"cf_main_tile", "cf_main_tile3e", "cf_main_feqmod", "cf_main_vah", "cf_main_vah3", "cf_main_direct"} <= names
    # M0 audit: the direct-to-LDS helper (cf_math.h::glds16a) sets M0 inside an asm statement; no other instruction of a kernel that stages
    # this way may read M0 (a compiler-generated user would have been handed the helper's value or have had its own overwritten)
    staged = [k for k, v in d.items() if v["global_load_lds"] > 0]
    # the modified-equilibrium kernel's row walk (cf_feqmod.hip): the row mask with the unit threshold only (ROWS = 2) issues fewest instructions
    fq = {r: d["cf_main_feqmod:DIM3=1,OUTFLOW=1,MODE3=0,JT=8,R=7,ROWS=%d" % r]["valu_f64_instr_per_eval"] for r in (0, 1, 2)}
    assert fq[2] < fq[0] and fq[2] < fq[1]
    assert len(staged) > 50 and "cf_main_tile3e:CE=1,OUTFLOW=1,REG=1,JT=8,R=7,MODE=1" in staged
    for k in staged:
        assert d[k]["m0_other_users"] == [] and d[k]["m0_writes"] >= 1, (k, d[k]["m0_other_users"])
    # the instantiation a re-stated template rule got wrong (round 2): the Chapman-Enskog 8 x 31 tile shares a reciprocal among 4
    # evaluations (cf_kernels.hip), the 14-moment one among 8
    ce = d["cf_main_tile:CE=1,DIM3=0,OUTFLOW=1,REG=1,BARYON=0,JT=8,R=31"]
    m14 = d["cf_main_tile:CE=0,DIM3=0,OUTFLOW=1,REG=1,BARYON=0,JT=8,R=31"]
    assert ce["evals_per_rcp"] == 4 and m14["evals_per_rcp"] == 8 and ce["flop_per_eval"] > m14["flop_per_eval"]
    # BASELINE config 3's kernel
    k3 = d["cf_main_tile3e:CE=1,OUTFLOW=1,REG=1,JT=8,R=7,MODE=1"]
    assert k3["evals_in_loop"] == 56 and k3["evals_per_rcp"] == 4
    # round 5: spills are accounted for -- no scratch instruction inside the counted loop of any kernel a default path launches (one there sits on the
    # critical path of every unit; the ones outside the loop park lane constants in the prologue / epilogue, once per wave)
    defaults = ["cf_main_tile3e:CE=1,OUTFLOW=1,REG=1,JT=8,R=7,MODE=1", "cf_main_tile3e:CE=0,OUTFLOW=1,REG=1,JT=8,R=7,MODE=1",
                "cf_main_tile3e_baryon:CE=1,OUTFLOW=1,REG=1,JT=8,R=7,MODE=1", "cf_main_tile3e_baryon:CE=0,OUTFLOW=1,REG=1,JT=8,R=7,MODE=1",
                "cf_main_tile:CE=0,DIM3=0,OUTFLOW=1,REG=1,BARYON=0,JT=8,R=31", "cf_main_tile:CE=1,DIM3=0,OUTFLOW=1,REG=1,BARYON=0,JT=8,R=31",
                "cf_main_feqmod:DIM3=1,OUTFLOW=1,MODE3=0,JT=8,R=7,ROWS=2", "cf_main_feqmod:DIM3=1,OUTFLOW=1,MODE3=1,JT=8,R=7,ROWS=2",
                "cf_main_feqmod:DIM3=0,OUTFLOW=1,MODE3=0,JT=8,R=31,ROWS=3", "cf_main_vah3:DIM3=1,REG=1,JT=8,R=7", "cf_main_vah3:DIM3=0,REG=1,JT=8,R=31"]
    for k in defaults:
        assert d[k]["scratch_in_loop"] == 0, (k, d[k]["scratch_in_loop"])
        assert d[k]["vgprs"] <= 256 and d[k]["occupancy_waves_per_simd"] >= 2, k
    # cf_main_vah3 in 3+1D kept its 56th accumulator in a scratch slot until round 5 (max_j d_j now rides in an SGPR pair)
    assert d["cf_main_vah3:DIM3=1,REG=1,JT=8,R=7"]["scratch_in_kernel"] <= 4
    # the scalar-path experiment of round 5 (developer build): more instructions per evaluation than the kernel it was to replace -- the spilled gamma_j
    assert d["cf_main_tile3s:CE=1,OUTFLOW=1,REG=1,JT=8,R=7"]["valu_f64_instr_per_eval"] >= k3["valu_f64_instr_per_eval"]
