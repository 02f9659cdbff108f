cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in 0.5 0.0625; do
  rm -rf /tmp/p_cap
  LITCODER_AMD_FIT_OPTS="screen_panel_first=$f" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cap -- python3 $R/tools/resident_fit_loop.py 3 2>&1 | grep "fit 2"
  echo "== first=$f"; python3 $R/tools/kstats.py /tmp/p_cap 40 < /dev/null | cut -c1-150
done
