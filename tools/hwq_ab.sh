# A/B of the runtime's hardware-queue count (GPU_MAX_HW_QUEUES, default 4) on the cfg2 fits: host-to-host and resident
for rep in 1 2; do
for q in ${HWQ_LIST:-4 8 16}; do
  echo "== GPU_MAX_HW_QUEUES=$q rep $rep"
  GPU_MAX_HW_QUEUES=$q python3 tools/resident_fit_loop.py 8 2>&1 | grep -E "fit [4-7]" | tr '\n' ' '; echo
  GPU_MAX_HW_QUEUES=$q python3 tools/host_fit_loop.py 7 2>&1 | grep -E "fit [3-6]" | tr '\n' ' '; echo
done; done
