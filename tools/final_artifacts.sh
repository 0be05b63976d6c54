#!/bin/bash
# Everything profiles/ holds for a round, from ONE box in ONE call (boxes differ by up to 10 % on the MFMA kernels):
#   bash tools/final_artifacts.sh r02_final
# -> gpurun_out/TAG_bench_unet.json (the default command's line), TAG_default_cmd_kernel_stats.csv (rocprofv3 of that same
#    command) + TAG_roofline_agreement.txt, TAG_unet_kernel_stats.csv / _kernels_by_grid.txt (graph replays only),
#    TAG_pmc_traffic.json, bench_TAG_<model>.json, TAG_ab_step.txt
TAG=${1:-r02_final}
python bench.py > gpurun_out/${TAG}_bench_unet.json 2> gpurun_out/${TAG}_bench_unet.err
head -c 240 gpurun_out/${TAG}_bench_unet.json; echo
bash tools/prof_default.sh > gpurun_out/${TAG}_roofline_agreement.txt 2>&1
cp gpurun_out/prof_default_kernel_stats.csv gpurun_out/${TAG}_default_cmd_kernel_stats.csv
cp gpurun_out/prof_default_bench.json gpurun_out/${TAG}_default_cmd_bench_line.json
cat gpurun_out/${TAG}_roofline_agreement.txt
bash tools/profile_bench.sh ${TAG}_unet > gpurun_out/pb.log 2>&1
bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/pmc_t.log 2>&1
bash tools/bench_models.sh ${TAG} > gpurun_out/${TAG}_bench_models.txt 2>&1
cat gpurun_out/${TAG}_bench_models.txt
python tools/ab_step.py --rounds 2 > gpurun_out/${TAG}_ab_step.txt 2>&1
tail -3 gpurun_out/${TAG}_ab_step.txt
