set -e
R=$GRAFT_REPO_ROOT
val() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'])" $1 $2; }
for i in 1 2; do
  (cd $R/ab_old && python bench.py --steps 50 --warmup 10 --no-f32-mfma-line --no-cpu-baseline > $R/gpurun_out/ab_old_lc$i.log 2> $R/gpurun_out/ab_old_lc$i.err) && val $R/gpurun_out/ab_old_lc$i.log old_LC
  (cd $R && python bench.py --steps 50 --warmup 10 --no-f32-mfma-line --no-cpu-baseline > $R/gpurun_out/ab_new_lc$i.log 2> $R/gpurun_out/ab_new_lc$i.err) && val $R/gpurun_out/ab_new_lc$i.log new_LC
  (cd $R/ab_old && python bench.py --workload nusc_L --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/ab_old_l$i.log 2> $R/gpurun_out/ab_old_l$i.err) && val $R/gpurun_out/ab_old_l$i.log old_L
  (cd $R && python bench.py --workload nusc_L --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/ab_new_l$i.log 2> $R/gpurun_out/ab_new_l$i.err) && val $R/gpurun_out/ab_new_l$i.log new_L
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl2 -o l -- python3 $R/bench.py --workload nusc_L --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/tl2.log 2>&1
cd $R
python tools/frame_kernel_list.py gpurun_out/tl2 > gpurun_out/tl2_frame.txt
tail -1 gpurun_out/tl2_frame.txt
rm -rf gpurun_out/tl2
