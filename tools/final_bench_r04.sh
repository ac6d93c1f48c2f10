#!/bin/bash
# Round-4 closing bench lines (GPU box, through gpurun): one JSON line per configuration into gpurun_out/final/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python bench.py > $O/r04_bench.json 2> $O/bench.err || exit 1
python bench.py --workload config2 > $O/r04_bench_config2.json 2> $O/c2.err || exit 2
python bench.py --workload config5 > $O/r04_bench_config5.json 2> $O/c5.err || exit 3
python bench.py --workload config5-sampler > $O/r04_bench_sampler.json 2> $O/smp.err || exit 4
python bench.py --workload config5 --dimension 2 --no-cpu-baseline > $O/r04_bench_config5_dim2.json 2> $O/c5d2.err || exit 5
python bench.py --workload config2 --df-mode 2 --no-cpu-baseline > $O/r04_bench_config2_ce.json 2> $O/c2ce.err || exit 6
python bench.py --df-mode 4 > $O/r04_bench_config3_feqmod4.json 2> $O/fq4.err || exit 9
python bench.py --df-mode 3 --no-cpu-baseline > $O/r04_bench_config3_feqmod3.json 2> $O/fq3.err || exit 10
python bench.py --workload config2 --df-mode 4 > $O/r04_bench_config2_feqmod4.json 2> $O/c2fq4.err || exit 11
python bench.py --workload config2 --df-mode 3 --no-cpu-baseline > $O/r04_bench_config2_feqmod3.json 2> $O/c2fq3.err || exit 12
python bench.py --rehearse-comm --no-cpu-baseline > $O/r04_bench_rehearse_comm.json 2> $O/rc.err || exit 7
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --backend gloo > $O/r04_bench_gpus2_gloo_rehearsal.json 2> $O/g2.err || exit 8
echo done
