for plan in "h:1-;r:0|1-" "h:1-;r:0-" "h:1|2-;r:0|1|2-" "h:1|2|3|4;r:0|1|2|3|4" "h:1-;r:0|1|2-" "h:1|2-;r:0|1-"; do
  echo "PLAN $plan"; LITCODER_SHARD_PLAN="$plan" python tools/main_stream_events.py 80000 8 0 2>&1 | grep "fit (device\|between"
done
