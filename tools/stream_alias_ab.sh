# Which of the engine's streams should share a queue: LITCODER_AMD_STREAM_ALIAS x GPU_MAX_HW_QUEUES on the cfg2 fits.
# streams: 0 aux (hat matrices), 1 aux2 (refit systems), 2 comm, 3 aux3, 4 dl, 5 scales, 6 side, 7 refine; m = default stream
med() { grep -E "fit [2-9]" | sed -E 's/.*: ([0-9.]+) ms.*/\1/' | sort -n | awk '{a[NR]=$1} END{printf "%.1f (min %.1f)", a[int((NR+1)/2)], a[1]}'; }
for q in ${HWQ_LIST:-4 8}; do
while read -r alias; do
  r=$(GPU_MAX_HW_QUEUES=$q LITCODER_AMD_STREAM_ALIAS="$alias" python3 tools/resident_fit_loop.py 7 2>&1 | med)
  h=$(GPU_MAX_HW_QUEUES=$q LITCODER_AMD_STREAM_ALIAS="$alias" python3 tools/host_fit_loop.py 7 2>&1 | med)
  echo "q=$q alias='$alias': resident $r   host $h"
done <<LIST
${ALIASES:-
7=1
7=0
7=m
1=0
2=7,3=7,5=7,6=7
2=0,3=0,5=0,6=0}
LIST
done
