export TMPDIR=/tmp
for C in 128 256; do
 for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  T=$(echo $P | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/pmcg/c${C}_$T -o p -- python3 tools/prof_gemm.py 556800 576 $C 6 > /dev/null 2>&1
  echo "== Cout=$C $P"; python tools/pmc_summary.py gpurun_out/pmcg/c${C}_$T srf_conv1x1_nhwc_k
 done
done
