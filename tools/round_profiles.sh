#!/bin/bash
# Collects the measurements of a round on the GPU box into gpurun_out/<round>/ (copy what is to be judged into profiles/).
# usage (from the repo root, through gpurun):  bash tools/round_profiles.sh r05 [A|B|C|all]
#   A = bench lines + kernel traces, B = per-stage tables / host feed / training / rehearsal, C = PMC traffic (a gpurun call is limited
#   to 20 minutes: one part per call)
set -o pipefail
R=${1:-r05}
PART=${2:-all}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
part() { { [ "$PART" = all ] && [ "$1" != T ]; } || [ "$PART" = "$1" ]; }
run() { echo "== $*" >> $O/log.txt; "$@" >> $O/log.txt 2>&1; }
line() { tail -1 "$1" > "$2"; }

if part T; then   # the two kernel traces alone
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lc -o lc -- python3 bench.py --no-cpu-baseline --no-f32-mfma-line > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_under_rocprof_LC.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l -o l -- python3 bench.py --workload nusc_L --no-cpu-baseline > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_under_rocprof_nuscL.json
cp $O/prof_lc/lc_kernel_stats.csv $O/${R}_bench_nuscLC_np200_kernel_stats.csv
cp $O/prof_l/l_kernel_stats.csv $O/${R}_bench_nuscL_np200_kernel_stats.csv
rm -f $O/${R}_in_graph_summary.json
python tools/in_graph_summary.py $O/${R}_bench_nuscLC_np200_kernel_stats.csv nusc_LC 88 $O/${R}_in_graph_summary.json >> $O/log.txt 2>&1
rm -rf $O/prof_lc $O/prof_l $O/tmp.json
fi
if part A; then
# bench lines (one JSON line each)
python bench.py > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line.json
python bench.py --workload nusc_L > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_nuscL.json
python bench.py --np 900 --no-cpu-baseline > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_np900_LC.json
python bench.py --workload nusc_L --np 900 --no-cpu-baseline > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_np900_nuscL.json
python bench.py --img-precomputed --no-cpu-baseline > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_img_precomputed_LC.json
python bench.py --workload waymo_L --steps 30 --warmup 5 --no-cpu-baseline > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_waymoL.json
python bench.py --workload kitti_L --steps 30 --warmup 5 --no-cpu-baseline > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_kittiL.json
python bench.py --whole-frame off --no-cpu-baseline > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_LC_tail_graph_only.json
echo "bench lines done" >> $O/log.txt

# kernel traces of the same commands
# (--no-f32-mfma-line: the traced process must contain the frames of the headline route only, or the per-frame sums mix both GEMM routes)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lc -o lc -- python3 bench.py --no-cpu-baseline --no-f32-mfma-line > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_under_rocprof_LC.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l -o l -- python3 bench.py --workload nusc_L --no-cpu-baseline > $O/tmp.json 2>> $O/log.txt && line $O/tmp.json $O/${R}_bench_line_under_rocprof_nuscL.json
cp $O/prof_lc/lc_kernel_stats.csv $O/${R}_bench_nuscLC_np200_kernel_stats.csv
cp $O/prof_l/l_kernel_stats.csv $O/${R}_bench_nuscL_np200_kernel_stats.csv
# ms per frame of the roofline kernel families inside the timed graphs (bench.py quotes it as in_graph_ms_per_frame)
python tools/in_graph_summary.py $O/${R}_bench_nuscLC_np200_kernel_stats.csv nusc_LC 88 $O/${R}_in_graph_summary.json >> $O/log.txt 2>&1
rm -rf $O/prof_lc $O/prof_l $O/tmp.json
echo "kernel traces done" >> $O/log.txt
fi
if part B; then

# per-stage tables, host-fed rates, the memset-node test
python tools/stage_roofline.py --workload nusc_L --md $O/${R}_stage_roofline_nuscL.md >> $O/log.txt 2>&1
python tools/stage_roofline.py --workload nusc_LC --frames 5 --md $O/${R}_stage_roofline_nuscLC.md >> $O/log.txt 2>&1
python tools/stage_roofline.py --workload waymo_L --frames 5 --md $O/${R}_stage_roofline_waymoL.md >> $O/log.txt 2>&1
python tools/feed_bench.py --workload nusc_LC > $O/${R}_feed_from_host_nuscLC.json 2>> $O/log.txt
python tools/feed_bench.py --workload nusc_L --steps 100 > $O/${R}_feed_from_host_nuscL.json 2>> $O/log.txt
[ -x tools/micro/graph_memset.bin ] && tools/micro/graph_memset.bin > $O/${R}_graph_memset_node_test.txt 2>&1
# config 4 on one GPU: iterations/s + the steady-state kernel table of a step
python tools/train_bench.py --iters 8 --warmup 3 --kernel-table $O/${R}_train_step_kernels.md 2>> $O/log.txt | grep metric > $O/${R}_train_bench_LC_bs2_np900.json
# N = 2 rehearsal of bench.py's distributed branch on the one GPU (gloo; both ranks on device 0)
SRF_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --workload nusc_L --no-cpu-baseline 2>> $O/log.txt | tail -1 > $O/${R}_rehearsal_2ranks_sharing_one_gpu_nuscL.json
echo "tables done" >> $O/log.txt
fi
if part C; then

# HBM traffic of the roofline kernels (three --pmc passes per target)
python tools/measure_traffic.py >> $O/log.txt 2>&1
cp gpurun_out/traffic/${R}_pmc_*_traffic.json $O/ 2>/dev/null
fi
echo "part $PART done" >> $O/log.txt
tail -3 $O/log.txt
