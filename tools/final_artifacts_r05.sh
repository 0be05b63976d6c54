#!/bin/bash
# Round 5's profiles/ set, from ONE box per part (a gpurun call is capped at 20 minutes):
#   bash tools/final_artifacts_r05.sh a TAG   default bench line (with the stock-PyTorch yardstick), rocprofv3 --kernel-trace --stats of
#                                             the default command + roofline agreement, graph-replay kernel tables
#   bash tools/final_artifacts_r05.sh b TAG   PMC traffic (unet, swin_unet_v2), same-box A/B of the Engine switches
PART=$1; TAG=${2:-r05}
mkdir -p gpurun_out
if [ "$PART" = a ]; then
  python3 bench.py > gpurun_out/${TAG}_final_bench_unet.json 2> gpurun_out/${TAG}_final_bench_unet.err
  head -c 300 gpurun_out/${TAG}_final_bench_unet.json; echo
  bash tools/prof_default.sh > gpurun_out/${TAG}_roofline_agreement.txt 2>&1
  cp gpurun_out/prof_default_kernel_stats.csv gpurun_out/${TAG}_default_cmd_kernel_stats.csv
  cp gpurun_out/prof_default_bench.json gpurun_out/${TAG}_default_cmd_bench_line.json
  cat gpurun_out/${TAG}_roofline_agreement.txt
  bash tools/profile_bench.sh ${TAG}_unet > gpurun_out/pb.log 2>&1
  bash tools/profile_bench.sh ${TAG}_swin_unet_v2_256 --model swin_unet_v2 > gpurun_out/pb2.log 2>&1
  tail -n 2 gpurun_out/pb.log; tail -n 2 gpurun_out/pb2.log
else
  bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_traffic.json --second-steps 0 > gpurun_out/pmc_t.log 2>&1
  tail -3 gpurun_out/pmc_t.log
  bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_traffic_swin_unet_v2_256.json --model swin_unet_v2 --second-steps 0 > gpurun_out/pmc_t2.log 2>&1
  tail -3 gpurun_out/pmc_t2.log
  python3 tools/ab_step.py --rounds 2 > gpurun_out/${TAG}_ab_step.txt 2>&1
  grep best gpurun_out/${TAG}_ab_step.txt
fi
