# Regenerates the round artefacts kept under profiles/ (run on the GPU box: gpurun -- bash tools/refresh_profiles.sh [tag]);
# outputs land in gpurun_out/<tag>_final/ and are copied into profiles/ by hand (see profiles/README.md).
set -x
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_final
mkdir -p $O
python3 $R/bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  D=/tmp/pmc_$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 $R/tools/resident_fit_loop.py 2 > /dev/null 2>&1
done
python3 $R/tools/parse_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES "k_sweep_f16x3<true" $O/sweep_fused_pmc.json > /dev/null
python3 $R/tools/parse_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES "k_sweep_f16x3<false, false, true, true" $O/sweep_series_pmc.json > /dev/null
python3 $R/tools/parse_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES "k_sweep_f16x3<false, false, false, false" $O/sweep_plain_pmc.json > /dev/null
# round 6: clock / matrix-pipe busy / bytes leaving L2 of the sweeps' full-width launches in one table (screening pass and refit)
python3 $R/tools/pmc_kernel_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES -- "k_sweep_f16x3<true" "k_sweep_f16x3<false, false, true, true" "k_sweep_f16x3<false, false, false, false, false, false>" "k_sweep_f16x3<false, false, false, false, true" > $O/sweep_kernels_pmc.json 2>&1
for K in k_mm64q k_lstep k_bstep k_potrf_diag; do
  python3 $R/tools/parse_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES "$K" $O/chol_${K}_pmc.json > /dev/null
done
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_res -- python3 $R/tools/resident_fit_loop.py 3 > $O/resident_fits.txt 2>&1
python3 $R/tools/queue_timeline.py $(find /tmp/p_res -name "*kernel_trace.csv" | head -1) 30 > $O/resident_queue_timeline.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_host -- python3 $R/tools/host_fit_loop.py 3 > $O/host_fits.txt 2>&1
python3 $R/tools/queue_timeline.py $(find /tmp/p_host -name "*kernel_trace.csv" | head -1) 30 > $O/host_queue_timeline.txt 2>&1
python3 $R/tools/gpu_kernel_bench.py sweep sweep16 stamps plain16 series lanczos chol hbm preproc > $O/kernel_microbench.txt 2>&1
python3 $R/tools/chol_ab.py > $O/chol_fused_ab.txt 2>&1
python3 $R/tools/scaling_model.py cfg2 cfg4 cfg5 cfg3 > $O/scaling_model.json 2> $O/scaling_model.err
python3 $R/tools/outlier_ab.py 8 > $O/outlier_ab.txt 2>&1
python3 $R/tools/ab_fits.py cfg3 8 default lanczos_dense=0 lanczos_tol=1e-6 > $O/ab_cfg3_lanczos.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_cfg3 -- python3 $R/tools/cfg3_fit_loop.py 3 > $O/cfg3_fits.txt 2>&1
python3 $R/tools/queue_timeline.py $(find /tmp/p_cfg3 -name "*kernel_trace.csv" | head -1) 200 > $O/cfg3_queue_timeline.txt 2>&1
python3 $R/tools/cfg3_probe.py 80000 h2d pipe fit > $O/cfg3_probe.txt 2>&1
python3 $R/tools/upload_probe.py > $O/upload_probe.txt 2>&1
python3 $R/tools/upload_probe.py 80000 busy 2>&1 | tail -4 > $O/upload_probe_beside_mfma.txt
python3 $R/tools/upload_probe.py 80000 busy hbm 2>&1 | tail -4 > $O/upload_probe_beside_hbm_passes.txt
python3 $R/tools/stream_queue_probe.py > $O/stream_queue_probe.txt 2>&1
python3 $R/tools/host_path_timeline.py 80000 cfg3 > $O/cfg3_host_path_timeline.txt 2>&1
python3 $R/tools/main_stream_events.py 80000 8 0 > $O/device_timeline_rank0_of_8.txt 2>&1
python3 $R/tools/main_stream_events.py 80000 > $O/device_timeline_1gpu.txt 2>&1
python3 $R/tools/host_path_timeline.py > $O/host_path_timeline.txt 2>&1
python3 $R/tools/other_configs.py > $O/other_configs.txt 2>&1
python3 $R/tools/kstats.py /tmp/prof_stats 60 > $O/bench_kernel_stats.txt 2>&1 < /dev/null
python3 $R/tools/screen_probe.py --reps 3 > $O/screen_probe_cfg2.txt 2>&1
ls -la $O
