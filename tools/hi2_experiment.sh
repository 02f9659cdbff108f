# Round 6, VERDICT r5 #1: the structure experiments on the dominant kernel, with counters.  Three forms of the inner CV's fused
# score sweep at cfg2, each under three counter passes of two resident fits:
#   three_mfma      FitOptions(screen_inner=0)            k_sweep_f16x3<score>: 3 MFMAs per product, 8 waves, 256 x 256 tiles
#   hi2_one_wg      FitOptions(screen_two_workgroups=0)   k_sweep_f16x3<score, HI2>: 1 MFMA per product, two K-tiles per barrier
#   hi2_two_wg      default                               k_sweep_hi2: the same on 256 x 128 tiles, two 4-wave workgroups per CU
# run on the GPU box:  gpurun -- bash tools/hi2_experiment.sh ; output: gpurun_out/hi2_experiment.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hi2_experiment.txt
: > $O
for V in three_mfma hi2_one_wg hi2_two_wg; do
  case $V in
    three_mfma) export FIT_OPTS="screen_inner=0";;
    hi2_one_wg) export FIT_OPTS="screen_two_workgroups=0";;
    hi2_two_wg) unset FIT_OPTS;;
  esac
  DIRS=""
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    D=/tmp/hi2_${V}_$(echo $C | cut -d' ' -f1)
    rm -rf $D
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 $R/tools/resident_fit_loop.py 2 > /dev/null 2>&1
    DIRS="$DIRS $D"
  done
  echo "=== $V (FIT_OPTS=${FIT_OPTS:-default}) ===" >> $O
  python3 $R/tools/pmc_kernel_summary.py $DIRS -- "k_sweep_f16x3<true" "k_sweep_hi2<true" "k_sweep_f16x3<false, false, true, true" "k_sweep_hi2<false" >> $O 2>&1
  echo "--- un-profiled, interleaved timing follows at the end ---" >> $O
done
unset FIT_OPTS
for rep in 1 2; do
  for V in "screen_inner=0" "screen_two_workgroups=0" ""; do
    FIT_OPTS=$V python3 $R/tools/resident_fit_loop.py 4 2>/dev/null | tail -3 | tr '\n' ' ' >> $O
    echo " <- FIT_OPTS='$V'" >> $O
  done
done
