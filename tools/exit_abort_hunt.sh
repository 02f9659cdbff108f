# An abort at process exit ("terminate called without an active exception") was seen ONCE in the unsharded worker of
# tools/fuzz_shards.py (seed 105): the same worker again and again, counting how often it dies.   bash tools/exit_abort_hunt.sh [runs] [seed]
N=${1:-12}; SEED=${2:-105}
T=$(mktemp -d /tmp/exit_hunt_XXXX)
python3 - "$T" <<'PY'
import sys, re
src = open("tools/fuzz_shards.py").read()
w = src[src.index("WORKER = r'''") + len("WORKER = r'''"):]
w = w[:w.index("'''")]
open(sys.argv[1] + "/worker.py", "w").write(w)
PY
bad=0
for i in $(seq 1 $N); do
  PYTHONFAULTHANDLER=1 MASTER_ADDR=127.0.0.1 python3 $T/worker.py $PWD none $T 30 $SEED > $T/out_$i.txt 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "run $i: rc=$rc"; tail -25 $T/out_$i.txt | cut -c1-200; fi
done
echo "$bad of $N runs died"
