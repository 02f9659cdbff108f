# A quick look at a resident and a host-to-host fit (queue timelines, kernel stats): gpurun -- bash tools/quick_prof.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quick
mkdir -p $O
rm -rf /tmp/p_res /tmp/p_host
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_res -- python3 $R/tools/resident_fit_loop.py 3 > $O/resident_fits.txt 2>&1
python3 $R/tools/queue_timeline.py $(find /tmp/p_res -name "*kernel_trace.csv" | head -1) 30 > $O/resident_queue_timeline.txt 2>&1
python3 $R/tools/queue_timeline.py $(find /tmp/p_res -name "*kernel_trace.csv" | head -1) 60 all > $O/resident_queue_timeline_all.txt 2>&1
python3 $R/tools/kstats.py /tmp/p_res 45 > $O/resident_kernel_stats.txt 2>&1 < /dev/null
if [ "$1" != "resident" ]; then
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_host -- python3 $R/tools/host_fit_loop.py 3 > $O/host_fits.txt 2>&1
python3 $R/tools/queue_timeline.py $(find /tmp/p_host -name "*kernel_trace.csv" | head -1) 30 > $O/host_queue_timeline.txt 2>&1
fi
