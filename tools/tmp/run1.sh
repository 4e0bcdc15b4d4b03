set -e
python -m pytest tests/test_gpu_glue.py tests/test_gpu_decoder.py tests/test_gpu_round3.py tests/test_gpu_integration.py tests/test_gpu_voxelize.py tests/test_gpu_spconv.py -x -q > gpurun_out/t_glue.log 2>&1 || { tail -60 gpurun_out/t_glue.log; exit 1; }
tail -3 gpurun_out/t_glue.log
python bench.py --workload nusc_L --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bL_glue.log 2> gpurun_out/bL_glue.err
tail -c 700 gpurun_out/bL_glue.log
python bench.py --steps 50 --warmup 10 --no-f32-mfma-line --no-cpu-baseline > gpurun_out/bLC_glue.log 2> gpurun_out/bLC_glue.err
tail -c 500 gpurun_out/bLC_glue.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload nusc_L --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/tl2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/frame_kernel_list.py gpurun_out/tl2 > gpurun_out/tl2_frame.txt
tail -2 gpurun_out/tl2_frame.txt
rm -rf gpurun_out/tl2
