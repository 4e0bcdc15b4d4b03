set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl2 -o l -- python3 $R/bench.py --workload nusc_L --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/tl2.log 2>&1
cd $R
python tools/frame_kernel_list.py gpurun_out/tl2 > gpurun_out/tl2_frame.txt
tail -1 gpurun_out/tl2_frame.txt
rm -rf gpurun_out/tl2
