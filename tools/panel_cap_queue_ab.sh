med() { grep -E "fit [2-9]" | sed -E 's/.*: ([0-9.]+) ms.*/\1/' | sort -n | awk '{a[NR]=$1} END{printf "%.1f (min %.1f)", a[int((NR+1)/2)], a[1]}'; }
for q in 8 4; do for f in 0.5 0.0625 0.02; do for al in "" "7=1"; do
  r=$(GPU_MAX_HW_QUEUES=$q LITCODER_AMD_STREAM_ALIAS="$al" LITCODER_AMD_FIT_OPTS="screen_panel_first=$f" python3 tools/resident_fit_loop.py 7 2>&1 | med)
  h=$(GPU_MAX_HW_QUEUES=$q LITCODER_AMD_STREAM_ALIAS="$al" LITCODER_AMD_FIT_OPTS="screen_panel_first=$f" python3 tools/host_fit_loop.py 7 2>&1 | med)
  echo "q=$q first=$f alias='$al': resident $r   host $h"
done; done; done
