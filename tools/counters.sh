#!/bin/bash
# SQ counter tables of the backward kernels and of fused SSIM -> gpurun_out/<round>_pmc_*.txt (GPU box).   bash tools/counters.sh r06
R=${1:-r06}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/pmc_table.sh k_render_bwd_tile tools/bench_stress.py > gpurun_out/${R}_pmc_bwd_tile_stress.txt 2>&1
ONE_CALL=0 WL=h36m STEPS=30 bash tools/pmc_table.sh k_render_bwd_wave tools/one_call_step.py > gpurun_out/${R}_pmc_bwd_wave_h36m.txt 2>&1
ONE_CALL=0 WL=panoptic STEPS=12 bash tools/pmc_table.sh k_render_bwd_wave tools/one_call_step.py > gpurun_out/${R}_pmc_bwd_wave_panoptic.txt 2>&1
bash tools/pmc_table.sh ssim tools/ssim_one.py fwd 6 5,1,1500,1500 > gpurun_out/${R}_pmc_ssim_fwd.txt 2>&1
bash tools/pmc_table.sh ssim tools/ssim_one.py train 6 5,1,1500,1500 > gpurun_out/${R}_pmc_ssim_train.txt 2>&1
tail -12 gpurun_out/${R}_pmc_bwd_tile_stress.txt
