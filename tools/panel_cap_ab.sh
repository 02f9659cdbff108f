for rep in 1 2; do
for f in 0.5 0.0625 0.15; do
  echo "== first=$f rep $rep"
  LITCODER_AMD_FIT_OPTS="screen_panel_first=$f" python3 tools/resident_fit_loop.py 8 2>&1 | grep -E "fit [4-7]" | tr '\n' ' '; echo
  LITCODER_AMD_FIT_OPTS="screen_panel_first=$f" python3 tools/host_fit_loop.py 6 2>&1 | grep -E "fit [2-5]" | tr '\n' ' '; echo
done; done
