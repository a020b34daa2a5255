#!/bin/bash
# usage: power_sample.sh <out.txt> <command...>  -- samples socket power and clocks (rocm-smi, ~5 Hz) while the command runs
out=$1; shift
( while true; do rocm-smi -P -g -M --csv 2>/dev/null | tr '\n' ' '; echo; sleep 0.2; done > "$out" ) &
sampler=$!
"$@"
rc=$?
kill $sampler 2>/dev/null
exit $rc
