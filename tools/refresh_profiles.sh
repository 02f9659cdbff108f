# Regenerates the round artifacts kept under profiles/ (run on the GPU box: gpurun -- bash tools/refresh_profiles.sh);
# outputs land in gpurun_out/r01_final/ and are copied into profiles/ by hand.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01_final
mkdir -p $O
python3 $R/bench.py --steps 3 --warmup 1 > $O/bench_v13.json 2> $O/bench_v13.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof_v13.json 2> /dev/null
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_v13.csv
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  D=/tmp/pmc_$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
python3 $R/tools/parse_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES "k_sweep_f16x3<true" $O/sweep_fused_pmc.json > /dev/null
python3 $R/tools/parse_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES "k_sweep_f16x3<false" $O/sweep_plain_pmc.json > /dev/null
python3 $R/tools/gpu_kernel_bench.py sweep sweep16 stamps plain16 series lanczos chol hbm > $O/kernel_microbench_v13.txt 2>&1
python3 $R/tools/overlap_probe.py >> $O/kernel_microbench_v13.txt 2>&1
python3 $R/tools/vendor_dgemm_probe.py > $O/fp64_rate_probe.txt 2>&1
$R/tools/bin/mfma_f64_rate >> $O/fp64_rate_probe.txt 2>&1
ls -la $O
