# The order in which the engine's streams get their hardware queues (LITCODER_AMD_STREAM_ORDER), and the two orders a process
# falls into by itself (first fit host to host / first fit resident): cfg2 host-to-host and resident fits in ONE process each.
med() { grep -E "fit [2-9]" | sed -E 's/.*: ([0-9.]+) ms.*/\1/' | sort -n | awk '{a[NR]=$1} END{printf "%.1f", a[int((NR+1)/2)]}'; }
run() { echo "order '$1': host first -> $(LITCODER_AMD_STREAM_ORDER="$1" python3 tools/host_then_resident.py host 2>&1 | tr '\n' ' ')"; echo "            resident first -> $(LITCODER_AMD_STREAM_ORDER="$1" python3 tools/host_then_resident.py resident 2>&1 | tr '\n' ' ')"; }
while read -r o; do run "$o"; done <<LIST
${ORDERS:-
0,1,7,2,3,4,5,6,u
u,5,0,1,7,2,4,3,6
0,1,7,u,5,4,2,3,6
u,0,1,7,5,4,2,3,6}
LIST
