#!/bin/bash
# samples the GPU clocks / power while a command runs: clock_watch.sh <out> -- cmd...
out=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|fclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > "$out" &
W=$!
"$@"
kill $W
