#!/bin/bash
# Run a command on the GPU box under a wall-clock limit AND a resident-memory watchdog: a host-side runaway (a Python list of millions
# of records) must end the command, not the box.   tools/guard.sh <seconds> <max RSS in GB> <command ...>
limit=$1; maxgb=$2; shift 2
setsid "$@" &
pid=$!
( t=0
  while kill -0 $pid 2>/dev/null; do
    sleep 2; t=$((t + 2))
    rss=$(ps -o rss= --sid $pid 2>/dev/null | awk '{s += $1} END {print int(s / 1048576)}')
    if [ "${rss:-0}" -gt "$maxgb" ]; then echo "guard: resident memory ${rss} GB > ${maxgb} GB: killing" >&2; kill -9 -- -$pid 2>/dev/null; break; fi
    if [ "$t" -gt "$limit" ]; then echo "guard: ${limit} s exceeded: killing" >&2; kill -9 -- -$pid 2>/dev/null; break; fi
  done ) &
wait $pid
rc=$?
echo "guard: exit $rc, $(free -g | awk '/Mem:/ {print "host memory " $3 " of " $2 " GB used"}')" >&2
exit $rc
