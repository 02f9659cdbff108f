# The counters of the sweep kernel's full-width launches alone (the part of tools/refresh_profiles.sh that makes
# <tag>_final/sweep_kernels_pmc.json): gpurun -- bash tools/pmc_sweep_summary.sh [tag]
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_final
mkdir -p $O
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  D=/tmp/pmc_$(echo $C | cut -d' ' -f1)
  rm -rf $D
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 $R/tools/resident_fit_loop.py 2 > /dev/null 2>&1
done
python3 $R/tools/pmc_kernel_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES -- "k_sweep_f16x3<true" "k_sweep_f16x3<false, false, true, true" "k_sweep_f16x3<false, false, false, false, false, false>" "k_sweep_f16x3<false, false, false, false, true" > $O/sweep_kernels_pmc.json 2>&1
