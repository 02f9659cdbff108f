# Queue timelines of a resident and a host-to-host fit under the caller's environment: gpurun -- 'GPU_MAX_HW_QUEUES=8 bash tools/quick_prof_env.sh tag'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quick_$1
mkdir -p $O
rm -rf /tmp/p_res /tmp/p_host
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_res -- python3 $R/tools/resident_fit_loop.py 3 > $O/resident_fits.txt 2>&1
python3 $R/tools/queue_timeline.py $(find /tmp/p_res -name "*kernel_trace.csv" | head -1) 60 all > $O/resident_queue_timeline_all.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_host -- python3 $R/tools/host_fit_loop.py 3 > $O/host_fits.txt 2>&1
python3 $R/tools/queue_timeline.py $(find /tmp/p_host -name "*kernel_trace.csv" | head -1) 60 all > $O/host_queue_timeline_all.txt 2>&1
grep "fit 2" $O/resident_fits.txt $O/host_fits.txt
