#!/bin/bash
# round 5, closing run: the bench lines of every workload with the final build -> gpurun_out/r05/final_*.json (copied to profiles/r05_bench*.json)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
run() { local n=$1; shift; python bench.py "$@" > $O/final_$n.json 2> $O/final_$n.err || { echo "$n failed"; tail -5 $O/final_$n.err; exit 1; }; python - "$O/final_$n.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
rv = d.get("roofline_valu") or {}
print(sys.argv[1].split("final_")[1], "step %.2f ms" % d["ms_per_step"], "value %.4g %s" % (d["value"], d["unit"]), "kernel_ms", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["kernel_ms"].items() if not isinstance(v, dict)},
      "valu frac %.4f (at clock %s, %s GHz)" % (rv.get("frac", 0), ("%.4f" % rv["frac_at_shader_clock"]) if rv.get("frac_at_shader_clock") else None, ("%.3f" % rv["shader_clock_ghz"]) if rv.get("shader_clock_ghz") else None),
      "cpu", d.get("cpu_baseline", {}).get("value"), "gpu/cpu executed", d.get("gpu_over_cpu_executed"))
PY
}
run default --steps 20 --warmup 5
run config5 --workload config5 --steps 5 --warmup 1
run config2 --workload config2 --steps 20 --warmup 5
run config3_feqmod4 --df-mode 4 --steps 5 --warmup 1
run config3_baryon --include-baryon --steps 5 --warmup 1
run config3_df1 --df-mode 1 --steps 5 --warmup 1
run sampler --workload config5-sampler --steps 10 --warmup 2
